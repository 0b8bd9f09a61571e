/* fastx.c — FASTA/FASTQ reader, gz-transparent, with the record framing of klib's kseq as used by the
 * reference (src/kseq.h:184-224, ks_getuntil2 :93-141).  Written from the behaviour, not from the macros:
 *   - jump to the next '>' or '@' (anywhere), name = bytes up to the first isspace(), comment = rest of
 *     that line unless the name was ended by '\n';
 *   - sequence: for every following line whose first byte is not '>', '@' or '+': '\n' alone is skipped,
 *     otherwise the line is appended; after each appended line one trailing '\r' is dropped once more than
 *     one byte is held (:138);
 *   - '+' starts a FASTQ quality block: the rest of the '+' line is skipped, quality lines are appended
 *     until at least as many bytes as the sequence are held; different lengths -> -2. */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include "cli.h"

#define FX_BUF (1 << 20)

struct cli_fastx {
    gzFile fp;
    unsigned char *buf;
    int begin, end, is_eof;
    int last_char;
    const unsigned char *pre; /* bytes to be read before the stream's own (cli_fastx_open_prefixed) */
    size_t pre_left;
};

/* "-" is the standard input for `sdust` alone (src/sdust/sdust.c:194); telofind, fa2bed and seq hand their argument to gzopen() as it is
 * (src/find_telomere.c:96, src/assbed.c:92, src/seq.c:106): a file of that name, or their "Could not to open file" and exit status 1 */
int cli_dash_is_stdin = 0;

cli_fastx_t *cli_fastx_open(const char *path)
{
    gzFile fp = (cli_dash_is_stdin && !strcmp(path, "-")) ? gzdopen(fileno(stdin), "r") : gzopen(path, "r");
    if (!fp) return NULL;
    gzbuffer(fp, 1 << 18);
    cli_fastx_t *f = (cli_fastx_t *)cli_xmalloc(sizeof(*f));
    memset(f, 0, sizeof(*f));
    f->fp = fp;
    f->buf = (unsigned char *)cli_xmalloc(FX_BUF);
    return f;
}

cli_fastx_t *cli_fastx_open_prefixed(void *gz, const void *prefix, size_t n)
{
    cli_fastx_t *f = (cli_fastx_t *)cli_xmalloc(sizeof(*f));
    memset(f, 0, sizeof(*f));
    f->fp = (gzFile)gz;
    f->buf = (unsigned char *)cli_xmalloc(FX_BUF);
    f->pre = (const unsigned char *)prefix;
    f->pre_left = n;
    return f;
}

void cli_fastx_close(cli_fastx_t *f)
{
    if (!f) return;
    gzclose(f->fp);
    free(f->buf);
    free(f);
}

static int fx_fill(cli_fastx_t *f)
{
    if (f->pre_left) {
        const size_t k = f->pre_left < FX_BUF ? f->pre_left : FX_BUF;
        memcpy(f->buf, f->pre, k);
        f->pre += k;
        f->pre_left -= k;
        f->begin = 0;
        f->end = (int)k;
        return 1;
    }
    if (f->is_eof) return 0;
    f->begin = 0;
    f->end = gzread(f->fp, f->buf, FX_BUF);
    if (f->end < FX_BUF) f->is_eof = 1;
    if (f->end <= 0) {
        f->end = 0;
        return 0;
    }
    return 1;
}

static inline int fx_getc(cli_fastx_t *f)
{
    if (f->begin >= f->end && !fx_fill(f)) return -1;
    return f->buf[f->begin++];
}

static void str_reserve(cli_str_t *s, size_t extra)
{
    if (s->l + extra + 1 > s->m) {
        s->m = (s->l + extra + 1) * 2;
        if (s->m < 256) s->m = 256;
        s->s = (char *)cli_xrealloc(s->s, s->m);
    }
}

/* append bytes up to (not including) the delimiter; mode 0: isspace(), mode 1: '\n'.
 * returns the delimiter byte, or -1 at end of input */
static int fx_until(cli_fastx_t *f, int line_mode, cli_str_t *s)
{
    int got_any = 0;
    for (;;) {
        if (f->begin >= f->end) {
            if (!fx_fill(f)) break;
        }
        got_any = 1;
        int i = f->begin;
        if (line_mode) {
            unsigned char *nl = (unsigned char *)memchr(f->buf + i, '\n', (size_t)(f->end - i));
            i = nl ? (int)(nl - f->buf) : f->end;
        } else {
            while (i < f->end && !isspace(f->buf[i])) ++i;
        }
        str_reserve(s, (size_t)(i - f->begin));
        memcpy(s->s + s->l, f->buf + f->begin, (size_t)(i - f->begin));
        s->l += (size_t)(i - f->begin);
        f->begin = i + 1;
        if (i < f->end) {
            if (line_mode && s->l > 1 && s->s[s->l - 1] == '\r') --s->l;
            str_reserve(s, 0);
            s->s[s->l] = 0;
            return f->buf[i];
        }
    }
    if (line_mode && s->l > 1 && s->s[s->l - 1] == '\r') --s->l;
    str_reserve(s, 0);
    s->s[s->l] = 0;
    return got_any ? 0 : -1;
}

int64_t cli_fastx_read(cli_fastx_t *f, cli_str_t *name, cli_str_t *comment, cli_str_t *seq, cli_str_t *qual)
{
    int c;
    if (f->last_char == 0) {
        while ((c = fx_getc(f)) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        f->last_char = c;
    }
    name->l = comment->l = seq->l = qual->l = 0;
    str_reserve(name, 0);
    str_reserve(comment, 0);
    str_reserve(seq, 0);
    str_reserve(qual, 0);
    name->s[0] = comment->s[0] = seq->s[0] = qual->s[0] = 0;
    c = fx_until(f, 0, name);
    if (c == -1) return -1;
    if (c != '\n' && c != 0) fx_until(f, 1, comment);
    while ((c = fx_getc(f)) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        str_reserve(seq, 1);
        seq->s[seq->l++] = (char)c;
        fx_until(f, 1, seq);
    }
    if (c == '>' || c == '@') f->last_char = c;
    str_reserve(seq, 0);
    seq->s[seq->l] = 0;
    if (c != '+') {
        if (c == -1) f->last_char = 0;
        return (int64_t)seq->l;
    }
    while ((c = fx_getc(f)) != -1 && c != '\n') {}
    if (c == -1) return -2;
    while (fx_until(f, 1, qual) >= 0 && qual->l < seq->l) {}
    f->last_char = 0;
    if (seq->l != qual->l) return -2;
    return (int64_t)seq->l;
}

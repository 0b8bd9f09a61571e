/* telobreaks_main.c — `cornetto telobreaks <lens_file> <sdust_file> <telomere_file>`: drop-in for
 * src/telomere_breaks.c:47-172 of the reference.  The three text files are read here with the reference's own
 * framing (fgets into LINE_MAX bytes, the leading white-space separated fields of every line, :62-66,:78-82,:93-97),
 * the bitset stage runs on the device (cornetto_telobreaks), and the lines come out in the bucket order of the
 * reference's khash table (cornetto_khash_str_order), :133-148.
 *
 * Where the reference has undefined behaviour this program stops with a message instead: a non-blank line that
 * lacks fields (sscanf leaves the variables of the previous line in place), coordinates outside the contig (the
 * bitset is indexed unchecked).  Blank lines are skipped. */
#include <ctype.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include "cli.h"

#define TB_LINE_MAX 2048 /* LINE_MAX of src/telomere_breaks.c:11-13 (glibc defines it as 2048) */

typedef struct {
    char **key;      /* distinct names (first spelling) */
    int32_t *id;     /* open addressing: id + 1, 0 = empty */
    uint32_t cap;
} tb_map_t;

static uint32_t tb_hash(const char *s)
{
    uint32_t h = 2166136261u;
    for (; *s; ++s) h = (h ^ (unsigned char)*s) * 16777619u;
    return h;
}

static void tb_map_build(tb_map_t *m, char **names, int32_t n_ids)
{
    m->cap = 16;
    while (m->cap < (uint32_t)n_ids * 2u + 2u) m->cap <<= 1;
    m->id = (int32_t *)cli_xmalloc(m->cap * sizeof(int32_t));
    memset(m->id, 0, m->cap * sizeof(int32_t));
    m->key = names;
    for (int32_t i = 0; i < n_ids; ++i) {
        uint32_t b = tb_hash(names[i]) & (m->cap - 1);
        while (m->id[b]) b = (b + 1) & (m->cap - 1);
        m->id[b] = i + 1;
    }
}

static int32_t tb_map_get(const tb_map_t *m, const char *name)
{
    uint32_t b = tb_hash(name) & (m->cap - 1);
    while (m->id[b]) {
        if (strcmp(m->key[m->id[b] - 1], name) == 0) return m->id[b] - 1;
        b = (b + 1) & (m->cap - 1);
    }
    return -1;
}

/* next white-space separated token of *p (sscanf's %s / the skipping before %d); NULL at the end of the line */
static char *tb_token(char **p)
{
    char *s = *p;
    while (*s && isspace((unsigned char)*s)) ++s;
    if (!*s) return NULL;
    char *e = s;
    while (*e && !isspace((unsigned char)*e)) ++e;
    if (*e) *e++ = '\0';
    *p = e;
    return s;
}

/* %d of sscanf: optional sign, decimal digits; the token may carry trailing garbage, which %d leaves unread */
static int tb_int(const char *tok, int *v)
{
    if (!tok) return 0;
    char *end;
    errno = 0;
    long x = strtol(tok, &end, 10);
    if (end == tok) return 0;
    *v = (int)x;
    return 1;
}

static FILE *tb_open(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) {                                               /* F_CHK, src/error.h:114-119 */
        CLI_ERROR("Failed to open %s : %s.", path, strerror(errno));
        exit(EXIT_FAILURE);
    }
    return f;
}

int telomere_breaks_main(int argc, char *argv[])
{
    if (argc < 4) {
        fprintf(stderr, "Usage: telobreaks <lens_file> <sdust_file> <telomere_file>\n");   /* :49 */
        return EXIT_FAILURE;
    }
    char line[TB_LINE_MAX];

    /* ---- lens file: name length (:62-75) ---- */
    char **names = NULL;
    int32_t *lens = NULL, n = 0, cap = 0;
    FILE *f = tb_open(argv[1]);
    while (fgets(line, sizeof(line), f)) {
        char *p = line, *nm = tb_token(&p);
        int len;
        if (!nm) continue;
        if (!tb_int(tb_token(&p), &len) || len < 0) {
            CLI_ERROR("%s: no contig length after '%s'", argv[1], nm);
            exit(EXIT_FAILURE);
        }
        if (n == cap) {
            cap = cap ? cap * 2 : 1024;
            names = (char **)cli_xrealloc(names, (size_t)cap * sizeof(char *));
            lens = (int32_t *)cli_xrealloc(lens, (size_t)cap * sizeof(int32_t));
        }
        names[n] = cli_xstrdup(nm);
        lens[n++] = len;
    }
    fclose(f);

    /* the reference's three khash maps get the same keys in the same order: one table decides the print order */
    int32_t *slot = (int32_t *)cli_xmalloc(((size_t)n + 1) * sizeof(int32_t));
    int32_t *order = (int32_t *)cli_xmalloc(((size_t)n + 1) * sizeof(int32_t));
    const int32_t n_ids = cornetto_khash_str_order((const char *const *)names, n, slot, order);
    if (n_ids < 0) {
        CLI_ERROR("contig name table failed");
        exit(EXIT_FAILURE);
    }
    char **id_name = (char **)cli_xmalloc(((size_t)n_ids + 1) * sizeof(char *));
    int32_t *id_len = (int32_t *)cli_xmalloc(((size_t)n_ids + 1) * sizeof(int32_t));
    for (int32_t i = 0; i < n_ids; ++i) id_name[i] = NULL;
    for (int32_t i = 0; i < n; ++i) {
        if (!id_name[slot[i]]) id_name[slot[i]] = names[i];      /* kh_put keeps the first key, src/khash.h:345 */
        id_len[slot[i]] = lens[i];                               /* the value is replaced, :69 */
    }
    tb_map_t map;
    tb_map_build(&map, id_name, n_ids);

    /* ---- sdust file: name start end (:78-90) ---- */
    cornetto_ivl_t *sd = NULL;
    int64_t n_sd = 0, cap_sd = 0;
    f = tb_open(argv[2]);
    while (fgets(line, sizeof(line), f)) {
        char *p = line, *nm = tb_token(&p);
        int a, b;
        if (!nm) continue;
        if (!tb_int(tb_token(&p), &a) || !tb_int(tb_token(&p), &b)) {
            CLI_ERROR("%s: a line of '%s' lacks start / end", argv[2], nm);
            exit(EXIT_FAILURE);
        }
        const int32_t id = tb_map_get(&map, nm);
        if (id < 0) continue;                                    /* :83 */
        if (n_sd == cap_sd) {
            cap_sd = cap_sd ? cap_sd * 2 : 1 << 16;
            sd = (cornetto_ivl_t *)cli_xrealloc(sd, (size_t)cap_sd * sizeof(*sd));
        }
        sd[n_sd].ctg = id; sd[n_sd].start = a; sd[n_sd].finish = b;
        ++n_sd;
    }
    fclose(f);

    /* ---- telomere file: name len strand start end matched (:93-97) ---- */
    cornetto_telrow_t *tel = NULL;
    int64_t n_tel = 0, cap_tel = 0;
    f = tb_open(argv[3]);
    while (fgets(line, sizeof(line), f)) {
        char *p = line, *nm = tb_token(&p);
        int skip, a, b, m;
        if (!nm) continue;
        if (!tb_int(tb_token(&p), &skip) || !tb_int(tb_token(&p), &skip) || !tb_int(tb_token(&p), &a) || !tb_int(tb_token(&p), &b) ||
            !tb_int(tb_token(&p), &m)) {
            CLI_ERROR("%s: a line of '%s' has fewer than six fields", argv[3], nm);
            exit(EXIT_FAILURE);
        }
        const int32_t id = tb_map_get(&map, nm);
        if (id < 0 || m < 24) continue;                          /* :98, :100 */
        if (n_tel == cap_tel) {
            cap_tel = cap_tel ? cap_tel * 2 : 1 << 16;
            tel = (cornetto_telrow_t *)cli_xrealloc(tel, (size_t)cap_tel * sizeof(*tel));
        }
        tel[n_tel].ctg = id; tel[n_tel].start = a; tel[n_tel].end = b; tel[n_tel].matched = m;
        ++n_tel;
    }
    fclose(f);

    /* ---- bitset stage on the device, :79-128 ---- */
    cornetto_ivl_t *runs = NULL;
    int64_t n_runs = 0;
    if (n_ids > 0 && n_sd > 0 && n_tel > 0 && cli_host_mode()) {
        if (cli_host_telobreaks(id_len, n_ids, sd, n_sd, tel, n_tel, &runs, &n_runs) != 0) {
            CLI_ERROR("%s", "telobreaks failed: an interval or telomere row lies outside its contig");
            exit(EXIT_FAILURE);
        }
    } else if (n_ids > 0 && n_sd > 0 && n_tel > 0) {
        cornetto_accel_t *h = cli_accel_open();
        cli_accel_check(h, cornetto_telobreaks(h, id_len, n_ids, sd, n_sd, tel, n_tel, &runs, &n_runs), "telobreaks");
        /* (the handle lives until exit: the result may sit in its pinned pool) */
    }

    /* ---- print, contigs in khash bucket order, runs by position (:133-148) ---- */
    int64_t *first = (int64_t *)cli_xmalloc(((size_t)n_ids + 2) * sizeof(int64_t));
    for (int32_t i = 0; i <= n_ids; ++i) first[i] = 0;
    for (int64_t r = 0; r < n_runs; ++r) ++first[runs[r].ctg + 1];
    for (int32_t i = 0; i < n_ids; ++i) first[i + 1] += first[i];
    for (int32_t k = 0; k < n_ids; ++k) {
        const int32_t id = order[k];
        for (int64_t r = first[id]; r < first[id + 1]; ++r)
            printf("Found telomere positions %d to %d is a telomere in %s of length %d\n", runs[r].start, runs[r].finish, id_name[id], id_len[id]);   /* :142 */
    }
    return EXIT_SUCCESS;
}

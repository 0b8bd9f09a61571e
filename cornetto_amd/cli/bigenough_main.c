/* bigenough_main.c — `cornetto bigenough [options] <assembly.bed> <boring.bed>`; host only (a few hundred
 * lines of text).  Reference: src/bigenough_main.c:229-296 (read_bed_to_hashmap), :92-149 (update_covlen),
 * :152-227 (print_bigenough_bits), :328-392 (main).  Reproduces the reference's 32-bit int arithmetic:
 * reg_t fields are int (:56-60), so covlen accumulates modulo 2^32 and the threshold product
 * (end-start)*T wraps before the division by 100 (:206). */
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

#include "cli.h"

typedef struct {
    char *name;
    int32_t start, end, covlen;
} ctg_t;

typedef struct {
    ctg_t *a;
    int32_t n, cap;
    int32_t *slots; /* open addressing: index into a, or -1 */
    int32_t n_slots;
} map_t;

static uint32_t str_hash(const char *s)
{
    uint32_t h = 2166136261u;
    for (; *s; ++s) h = (h ^ (uint8_t)*s) * 16777619u;
    return h;
}

static void map_rehash(map_t *m, int32_t n_slots)
{
    free(m->slots);
    m->n_slots = n_slots;
    m->slots = (int32_t *)cli_xmalloc((size_t)n_slots * sizeof(int32_t));
    for (int32_t i = 0; i < n_slots; ++i) m->slots[i] = -1;
    for (int32_t i = 0; i < m->n; ++i) {
        uint32_t k = str_hash(m->a[i].name) & (uint32_t)(n_slots - 1);
        while (m->slots[k] >= 0) k = (k + 1) & (uint32_t)(n_slots - 1);
        m->slots[k] = i;
    }
}

static ctg_t *map_get(map_t *m, const char *name)
{
    if (!m->n_slots) return NULL;
    uint32_t k = str_hash(name) & (uint32_t)(m->n_slots - 1);
    while (m->slots[k] >= 0) {
        if (strcmp(m->a[m->slots[k]].name, name) == 0) return &m->a[m->slots[k]];
        k = (k + 1) & (uint32_t)(m->n_slots - 1);
    }
    return NULL;
}

static ctg_t *map_put(map_t *m, const char *name)
{
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 256;
        m->a = (ctg_t *)cli_xrealloc(m->a, (size_t)m->cap * sizeof(ctg_t));
    }
    if ((m->n + 1) * 2 > m->n_slots) map_rehash(m, m->n_slots ? m->n_slots * 2 : 512);
    m->a[m->n].name = cli_xstrdup(name);
    uint32_t k = str_hash(name) & (uint32_t)(m->n_slots - 1);
    while (m->slots[k] >= 0) k = (k + 1) & (uint32_t)(m->n_slots - 1);
    m->slots[k] = m->n;
    return &m->a[m->n++];
}

/* one BED line with the reference's checks (:175-196); ref is NUL-terminated inside line */
static void parse_bed_line(char *line, ssize_t len, const char *bedfile, int64_t line_no, char **ref, int64_t *beg, int64_t *end)
{
    (void)len;
    *beg = -1;
    *end = -1;
    char *p = line;
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f') ++p;
    int ret = 0;
    *ref = p;
    if (*p) {
        while (*p && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
        char *name_end = p;
        ret = 1;
        char *q;
        long v = strtol(p, &q, 10);   /* %ld skips leading white space itself */
        if (q != p && *p) {
            /* strtol accepted something: make sure digits were consumed */
            *beg = v;
            ret = 2;
            p = q;
            v = strtol(p, &q, 10);
            if (q != p) {
                *end = v;
                ret = 3;
            }
        }
        *name_end = 0;
    }
    if (ret != 3 || *end < *beg) {
        CLI_ERROR("Malformed bed entry at line %ld", (long)line_no);
        exit(EXIT_FAILURE);
    }
    if (*beg < 0 || *end < 0) {
        CLI_ERROR("Malformed bed entry at %s:%ld. Coordinates cannot be negative", bedfile, (long)line_no);
        exit(EXIT_FAILURE);
    }
    if (*beg >= *end) {
        CLI_ERROR("Malformed bed entry at %s:%ld. start must be smaller than end coordinate", bedfile, (long)line_no);
        exit(EXIT_FAILURE);
    }
}

static FILE *open_or_die(const char *path, const char *mode)
{
    FILE *fp = fopen(path, mode);
    if (!fp) {
        CLI_ERROR("Failed to open %s : No such file or directory.", path);
        exit(EXIT_FAILURE);
    }
    return fp;
}

int bigenough_main(int argc, char *argv[])
{
    static const struct option lo[] = {{"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},
                                       {"version", no_argument, 0, 'V'}, {"threshold", required_argument, 0, 'T'},
                                       {"readfish", required_argument, 0, 'r'}, {0, 0, 0, 0}};
    int threshold = 50; /* :87 */
    const char *outreadfish = NULL;
    FILE *fp_help = stderr;
    int c, li = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "T:v:r:hV", lo, &li)) >= 0) {
        if (c == 'T') {
            threshold = atoi(optarg);
            if (threshold < 0 || threshold > 100) {
                CLI_ERROR("Threshold should be between 0 and 100. You entered %d", threshold);
                exit(EXIT_FAILURE);
            }
        } else if (c == 'r') {
            outreadfish = optarg;
        } else if (c == 'v') {
            cli_log_level = atoi(optarg);
        } else if (c == 'V') {
            fprintf(stdout, "cornetto %s\n", CORNETTO_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        }
    }
    if (argc - optind != 2 || fp_help == stdout) {
        fprintf(fp_help, "Usage: cornetto bigenough [options] <assembly.bed> <boring.bed>\n");
        fprintf(fp_help, "   -T INT                     percentage threshold to consider as sufficient boring bits on a contig [%d]\n", threshold);
        fprintf(fp_help, "   -r FILE                    also output in readfish format to FILE\n");
        fprintf(fp_help, "   -v INT                     verbosity level [%d]\n", cli_log_level);
        fprintf(fp_help, "   -h                         help\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    const char *assbed = argv[optind], *boringbed = argv[optind + 1];
    map_t m;
    memset(&m, 0, sizeof(m));
    char *line = NULL;
    size_t lcap = 0;
    ssize_t ll;
    int64_t line_no = 0;
    uint64_t asslen = 0, boring_len = 0, panel_len = 0;

    /* read_bed_to_hashmap :229-296 */
    FILE *fp = open_or_die(assbed, "r");
    while ((ll = getline(&line, &lcap, fp)) != -1) {
        char *ref;
        int64_t beg, end;
        parse_bed_line(line, ll, assbed, line_no, &ref, &beg, &end);
        if (beg != 0) {
            CLI_ERROR("start coordinate should be 0 in the assembly chromosome bed. Not so at %s:%ld. ", assbed, (long)line_no);
            exit(EXIT_FAILURE);
        }
        if (map_get(&m, ref)) {
            CLI_ERROR("Contig '%s' is duplicated in %s", ref, assbed);
            exit(EXIT_FAILURE);
        }
        ctg_t *r = map_put(&m, ref);
        r->start = (int32_t)beg; /* int64 -> int, as the reference stores it */
        r->end = (int32_t)end;
        r->covlen = 0;
        asslen += (uint64_t)end;
        line_no++;
    }
    fclose(fp);

    /* update_covlen :92-149 */
    fp = open_or_die(boringbed, "r");
    line_no = 0;
    while ((ll = getline(&line, &lcap, fp)) != -1) {
        char *ref;
        int64_t beg, end;
        parse_bed_line(line, ll, boringbed, line_no, &ref, &beg, &end);
        ctg_t *r = map_get(&m, ref);
        if (!r) {
            CLI_ERROR("Contig '%s' in %s is not found in assembly bed file", ref, boringbed);
            exit(EXIT_FAILURE);
        }
        r->covlen = (int32_t)((uint32_t)r->covlen + (uint32_t)(end - beg)); /* int += int64, wraps */
        boring_len += (uint64_t)(end - beg);
        line_no++;
    }
    fclose(fp);

    /* print_bigenough_bits :152-227 */
    fp = open_or_die(boringbed, "r");
    FILE *outfp = outreadfish ? open_or_die(outreadfish, "w") : NULL;
    line_no = 0;
    while ((ll = getline(&line, &lcap, fp)) != -1) {
        char *ref;
        int64_t beg, end;
        parse_bed_line(line, ll, boringbed, line_no, &ref, &beg, &end);
        ctg_t *r = map_get(&m, ref);
        if (!r) {
            CLI_ERROR("Contig '%s' in %s is not found in assembly bed file", ref, boringbed);
            exit(EXIT_FAILURE);
        }
        const int32_t prod = (int32_t)((uint32_t)(r->end - r->start) * (uint32_t)threshold); /* :206, int wrap */
        if (r->covlen > prod / 100) {
            printf("%s\t%ld\t%ld\n", ref, (long)beg, (long)end);
            if (outfp) {
                fprintf(outfp, "%s,%ld,%ld,+\n", ref, (long)beg, (long)end);
                fprintf(outfp, "%s,%ld,%ld,-\n", ref, (long)beg, (long)end);
            }
            panel_len += (uint64_t)(end - beg);
        }
        line_no++;
    }
    fclose(fp);
    if (outfp) fclose(outfp);
    free(line);

    /* :317-321 */
    fprintf(stderr, "Total assembly length:\t%ld\t%.2f Gbases\n", (long)asslen, asslen / 1000000000.0);
    fprintf(stderr, "boring bits length before filtering:\t%ld\t%.2f Gbases\n", (long)boring_len, boring_len / 1000000000.0);
    fprintf(stderr, "Final panel length:\t%ld\t%.2f Gbases\n", (long)panel_len, panel_len / 1000000000.0);
    fprintf(stderr, "%% of panel length (over assembly):\t%.2f%%\n", (float)panel_len / (float)asslen * 100);
    fprintf(stderr, "%% of panel length (over human genome):\t%.2f%%\n", (float)panel_len / (float)3100000000 * 100);

    for (int32_t i = 0; i < m.n; ++i) free(m.a[i].name);
    free(m.a);
    free(m.slots);
    return 0;
}

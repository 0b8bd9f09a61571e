/* telowin_main.c — `cornetto telowin <in.telomere> <identity> [threshold]`; reference:
 * src/telomere_windows.c:45-86.  The TSV is parsed on the host exactly as the reference does
 * (sscanf of six whitespace-separated strings :67, atoi of columns 2,4,5, a new contig whenever the name
 * differs from the previous line :69-74); marking and the 1000/200 window scan run on the device. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cli.h"

int telomere_windows_main(int argc, char *argv[])
{
    if (argc < 3) { /* :46 */
        fprintf(stderr, "Usage: cornetto telowin <input_file> <identity> <threshold>\n");
        fprintf(stderr, "This program analyzes telomere windows in a genome assembly.\n");
        fprintf(stderr, "Example usage: cornetto telowin input.telomere 99.9 0.4\n");
        return EXIT_FAILURE;
    }
    double threshold = 0.4; /* :20 */
    if (argc == 4) threshold = atof(argv[3]); /* :48-50 */
    const double identity = atof(argv[2]) / 100;
    const double thr_adj = cornetto_telowin_threshold(threshold, atof(argv[2]));
    fprintf(stderr, "Given error rate of %.6f running with adjusted threshold of %.6f due to survival prob %.6f\n",
            identity, thr_adj, pow(identity, 6)); /* :55 */

    FILE *fp = fopen(argv[1], "r");
    if (!fp) {
        CLI_ERROR("Failed to open %s : No such file or directory.", argv[1]);
        exit(EXIT_FAILURE);
    }
    char **names = NULL;
    int32_t *lens = NULL;
    int32_t n_ctg = 0, cap_ctg = 0;
    cornetto_hit_t *hits = NULL;
    int64_t n_hits = 0, cap_hits = 0;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t ll;
    while ((ll = getline(&line, &lcap, fp)) != -1) {
        char *tok[6] = {"", "", "", "", "", ""};
        int nt = 0;
        char *p = line;
        while (nt < 6) { /* "%s %s %s %s %s %s" */
            while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f') ++p;
            if (!*p) break;
            tok[nt++] = p;
            while (*p && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
            if (*p) *p++ = 0;
        }
        if (nt == 0 && n_ctg == 0) continue; /* blank leading line: the reference would read garbage; skip */
        if (n_ctg == 0 || strcmp(tok[0], names[n_ctg - 1]) != 0) { /* :69-74 */
            if (n_ctg == cap_ctg) {
                cap_ctg = cap_ctg ? cap_ctg * 2 : 64;
                names = (char **)cli_xrealloc(names, (size_t)cap_ctg * sizeof(char *));
                lens = (int32_t *)cli_xrealloc(lens, (size_t)cap_ctg * sizeof(int32_t));
            }
            names[n_ctg] = cli_xstrdup(tok[0]);
            lens[n_ctg] = atoi(tok[1]);
            if (lens[n_ctg] < 0) {
                CLI_ERROR("negative contig length for %s", tok[0]);
                exit(EXIT_FAILURE);
            }
            n_ctg++;
        }
        const int start = atoi(tok[3]), end = atoi(tok[4]); /* :75-76 */
        if (start < end) {
            if (start < 0 || end > lens[n_ctg - 1]) { /* the reference writes out of bounds here */
                CLI_ERROR("hit %d-%d lies outside contig %s of length %d", start, end, names[n_ctg - 1], lens[n_ctg - 1]);
                exit(EXIT_FAILURE);
            }
            if (n_hits == cap_hits) {
                cap_hits = cap_hits ? cap_hits * 2 : 1024;
                hits = (cornetto_hit_t *)cli_xrealloc(hits, (size_t)cap_hits * sizeof(cornetto_hit_t));
            }
            hits[n_hits].ctg = n_ctg - 1;
            hits[n_hits].strand = atoi(tok[2]);
            hits[n_hits].start = start;
            hits[n_hits].end = end;
            n_hits++;
        }
    }
    free(line);
    fclose(fp);

    if (n_ctg > 0) {
        const int host = cli_host_mode();
        cornetto_accel_t *h = host ? NULL : cli_accel_open();
        cornetto_win_t *wins = NULL;
        int64_t n_wins = 0;
        if (host) cli_host_telowin(hits, n_hits, lens, n_ctg, thr_adj, &wins, &n_wins);
        else cli_accel_check(h, cornetto_telowin(h, hits, n_hits, lens, n_ctg, thr_adj, &wins, &n_wins), "telowin");
        for (int64_t i = 0; i < n_wins; ++i) { /* :38 */
            const int den = wins[i].end - wins[i].start;
            printf("Window\t%s\t%d\t%d\t%d\t%.3g\n", names[wins[i].ctg], lens[wins[i].ctg], wins[i].start, wins[i].end,
                   (double)wins[i].car / den);
        }
        if (host) free(wins);
        else cornetto_free(wins);
        if (h) cornetto_accel_close(h);
    }
    for (int32_t i = 0; i < n_ctg; ++i) free(names[i]);
    free(names);
    free(lens);
    free(hits);
    return EXIT_SUCCESS;
}

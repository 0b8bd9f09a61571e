/* depth_main.c — `cornetto depth reads.bam`: in the reference this sub-command parses its options and
 * does no work (the batch loop is commented out, src/depth_main.c:164-191): it prints zeroed statistics to
 * stderr and returns 0 without ever opening the BAM.  Kept as the same CLI-compatible stub. */
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

#include "cli.h"

int depth_main(int argc, char *argv[])
{
    static const struct option lo[] = {
        {"threads", required_argument, 0, 't'},   {"batchsize", required_argument, 0, 'K'},
        {"max-bytes", required_argument, 0, 'B'}, {"verbose", required_argument, 0, 'v'},
        {"help", no_argument, 0, 'h'},            {"version", no_argument, 0, 'V'},
        {"output", required_argument, 0, 'o'},    {"debug-break", required_argument, 0, 0},
        {"profile-cpu", required_argument, 0, 0}, {"accel", required_argument, 0, 0},
        {0, 0, 0, 0}};
    FILE *fp_help = stderr;
    int num_thread = 8, batch_size = 512;
    int c, li = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "t:B:K:v:o:hV", lo, &li)) >= 0) {
        if (c == 'K') {
            batch_size = atoi(optarg);
            if (batch_size < 1) {
                CLI_ERROR("Batch size should larger than 0. You entered %d", batch_size);
                exit(EXIT_FAILURE);
            }
        } else if (c == 't') {
            num_thread = atoi(optarg);
            if (num_thread < 1) {
                CLI_ERROR("Number of threads should larger than 0. You entered %d", num_thread);
                exit(EXIT_FAILURE);
            }
        } else if (c == 'v') {
            cli_log_level = atoi(optarg);
        } else if (c == 'V') {
            fprintf(stdout, "cornetto %s\n", CORNETTO_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        }
    }
    if (argc - optind != 1 || fp_help == stdout) {
        fprintf(fp_help, "Usage: cornetto depth reads.bam\n");
        fprintf(fp_help, "\nbasic options:\n");
        fprintf(fp_help, "   -t INT                     number of processing threads [%d]\n", num_thread);
        fprintf(fp_help, "   -K INT                     batch size (max number of reads loaded at once) [%d]\n", batch_size);
        fprintf(fp_help, "   -h                         help\n");
        fprintf(fp_help, "   --verbose INT              verbosity level [%d]\n", cli_log_level);
        fprintf(fp_help, "   --version                  print version\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    /* src/depth_main.c:196-207 */
    fprintf(stderr, "[%s] total entries: %ld", __func__, 0L);
    fprintf(stderr, "\n[%s] total bytes: %.1f M", __func__, 0.0);
    fprintf(stderr, "\n[%s] Data loading time: %.3f sec", __func__, 0.0);
    fprintf(stderr, "\n[%s] Data processing time: %.3f sec", __func__, 0.0);
    fprintf(stderr, "\n[%s] Data output time: %.3f sec", __func__, 0.0);
    fprintf(stderr, "\n");
    return 0;
}

/* main.c — `cornetto <command>` dispatcher; mirrors src/main.c:95-152 of the reference for the
 * panel-creation sub-commands (same names, same exit codes, same 3-line stderr footer).  Sub-commands of
 * the reference that are outside this path (fixasm, minidot, asmstats, nx, report, telocontigs)
 * are named in the usage text as not built here and exit with status 1; telobreaks
 * is built and dispatched below. */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cli.h"

static int print_usage(FILE *fp)
{
    fprintf(fp, "Usage: cornetto <command> [options]\n\n");
    fprintf(fp, "commands (MI355X build: panel-creation path only):\n");
    fprintf(fp, "   create panel:\n");
    fprintf(fp, "       noboringbits    print no boring bits in an assembly\n");
    fprintf(fp, "       boringbits      print boring bits in an assembly (deprecated)\n");
    fprintf(fp, "       bigenough       find contigs that have sufficient boring bits\n");
    fprintf(fp, "       depth           (stub, as in the reference)\n");
    fprintf(fp, "   telo:\n");
    fprintf(fp, "       telowin         analyse telomere windows in a fasta file\n");
    fprintf(fp, "       telofind        find telomere sequences in a fasta file\n");
    fprintf(fp, "       telobreaks      find telomere sequences inside low-complexity runs\n");
    fprintf(fp, "       sdust           symmetric DUST (https://github.com/lh3/sdust)\n");
    fprintf(fp, "   misc:\n");
    fprintf(fp, "       fa2bed          create a bed file with assembly contig lengths\n");
    fprintf(fp, "       seq             extract reads equal or longer than a threshold from a fastq\n");
    fprintf(fp, "\n");
    fprintf(fp, "       --help, -h      print this help message\n");
    fprintf(fp, "       --version, -V   print version information\n");
    return fp == stdout ? EXIT_SUCCESS : EXIT_FAILURE;
}

int main(int argc, char *argv[])
{
    double realtime0 = cli_realtime();
    int ret = 1;
    if (argc < 2) {
        return print_usage(stderr);
    } else if (strcmp(argv[1], "depth") == 0) {
        ret = depth_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "boringbits") == 0) {
        ret = boringbits_main(argc - 1, argv + 1, 1);
    } else if (strcmp(argv[1], "noboringbits") == 0) {
        ret = boringbits_main(argc - 1, argv + 1, 0);
    } else if (strcmp(argv[1], "telowin") == 0) {
        ret = telomere_windows_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "telobreaks") == 0) {
        ret = telomere_breaks_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "telofind") == 0) {
        ret = find_telomere_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "bigenough") == 0) {
        ret = bigenough_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "sdust") == 0) {
        ret = sdust_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "fa2bed") == 0) {
        ret = assbed_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "seq") == 0) {
        ret = seq_main(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "-V") == 0) {
        fprintf(stdout, "cornetto %s\n", CORNETTO_VERSION);
        exit(EXIT_SUCCESS);
    } else if (strcmp(argv[1], "--help") == 0 || strcmp(argv[1], "-h") == 0) {
        return print_usage(stdout);
    } else {
        fprintf(stderr, "[cornetto] Unrecognised command %s\n", argv[1]);
        return print_usage(stderr);
    }
    fprintf(stderr, "[%s] Version: %s (MI355X/HIP backend)\n", __func__, CORNETTO_VERSION);
    fprintf(stderr, "[%s] CMD:", __func__);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fprintf(stderr, "\n[%s] Real time: %.3f sec; CPU time: %.3f sec; Peak RAM: %.3f GB\n\n", __func__,
            cli_realtime() - realtime0, cli_cputime(), cli_peakrss() / 1024.0 / 1024.0 / 1024.0);
    /* everything is printed: leave without the orderly teardown of the HIP runtime and of gigabytes of pinned memory
     * (0.1-0.2 s that the kernel does faster) */
    fflush(NULL);
    _exit(ret);
}

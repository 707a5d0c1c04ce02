/* A plain C99 client of libspx (no Python, no C++): reads one problem from a binary file, runs the one-shot
 * EI grid and the log-likelihood through the C ABI exactly as include/spx.h declares it, writes the results.
 * Built and run by tests/test_gpu_multi.py::test_plain_c_client_of_the_abi (gcc, -lspx).
 *   file in : int64 N, D, M, H; double comp[N*D], vals[N], cand[M*D], hypers[H*(3+D)]
 *   file out: int64 best_idx; double best_val, mean[M], draws[M*H], lp[H]                              */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "spx.h"

static int die(const char* what)
{
    fprintf(stderr, "%s: %s\n", what, spx_last_error());
    return 1;
}

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: abi_client in.bin out.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t dims[4];
    if (fread(dims, sizeof(int64_t), 4, f) != 4) return 2;
    const int64_t N = dims[0], D = dims[1], M = dims[2], H = dims[3];
    double* comp = malloc(sizeof(double) * N * D);
    double* vals = malloc(sizeof(double) * N);
    double* cand = malloc(sizeof(double) * M * D);
    double* hyp = malloc(sizeof(double) * H * (3 + D));
    if (fread(comp, sizeof(double), N * D, f) != (size_t)(N * D) || fread(vals, sizeof(double), N, f) != (size_t)N ||
        fread(cand, sizeof(double), M * D, f) != (size_t)(M * D) ||
        fread(hyp, sizeof(double), H * (3 + D), f) != (size_t)(H * (3 + D)))
        return 2;
    fclose(f);

    if (spx_version() < 200) return die("library too old");
    spx_handle* h = NULL;
    if (spx_create(0, &h) != SPX_OK) return die("spx_create");
    double* mean = malloc(sizeof(double) * M);
    double* draws = malloc(sizeof(double) * M * H);
    double* lp = malloc(sizeof(double) * H);
    int64_t best_idx = -1;
    double best_val = 0.0;
    if (spx_ei_grid(h, comp, vals, N, (int32_t)D, cand, M, hyp, (int32_t)H, 0, mean, draws, &best_idx, &best_val) != SPX_OK)
        return die("spx_ei_grid");
    /* resident-data calls: the observations are still on the device */
    if (spx_set_hypers(h, hyp, (int32_t)H) != SPX_OK) return die("spx_set_hypers");
    if (spx_gp_logprob(h, lp) != SPX_OK) return die("spx_gp_logprob");
    /* an error is a return code and a message, never an exit() */
    if (spx_set_option(h, "no_such_option", 1) != SPX_ERR_ARG) return die("expected SPX_ERR_ARG");
    spx_destroy(h);

    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(&best_idx, sizeof best_idx, 1, f);
    fwrite(&best_val, sizeof best_val, 1, f);
    fwrite(mean, sizeof(double), M, f);
    fwrite(draws, sizeof(double), M * H, f);
    fwrite(lp, sizeof(double), H, f);
    fclose(f);
    printf("best %lld %.17g\n", (long long)best_idx, best_val);
    return 0;
}

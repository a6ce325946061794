#define _POSIX_C_SOURCE 200809L
/*
 * orz_oracle_cli.c -- file-to-file driver for the parity oracle (TEST INFRASTRUCTURE ONLY).
 * Usage: orz_oracle encode -l{0,1,2} IN OUT | orz_oracle decode IN OUT | orz_oracle time -l1 IN [reps]
 * `time` prints one JSON line with the single-thread encode throughput (cpu_baseline in bench.py).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "orz_oracle.h"

static uint8_t* slurp(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* p = (uint8_t*)malloc(sz > 0 ? (size_t)sz : 1);
    *n = fread(p, 1, (size_t)sz, f);
    fclose(f);
    return p;
}
static int spit(const char* path, const uint8_t* p, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    size_t w = fwrite(p, 1, n, f);
    fclose(f);
    return w == n ? 0 : -1;
}
static int level_cfg(const char* a, orc_lzcfg* c) { /* src/main.rs:97-102 */
    if (!strcmp(a, "-l0")) { c->match_depth = 5; c->lazy_match_depth1 = 3; c->lazy_match_depth2 = 2; return 0; }
    if (!strcmp(a, "-l1")) { c->match_depth = 15; c->lazy_match_depth1 = 9; c->lazy_match_depth2 = 6; return 0; }
    if (!strcmp(a, "-l2")) { c->match_depth = 45; c->lazy_match_depth1 = 27; c->lazy_match_depth2 = 18; return 0; }
    return -1;
}
static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char** argv) {
    orc_lzcfg cfg;
    if (argc == 5 && !strcmp(argv[1], "encode") && !level_cfg(argv[2], &cfg)) {
        size_t n, m;
        uint8_t *in = slurp(argv[3], &n), *out;
        if (!in || orc_encode_mem(in, n, &cfg, &out, &m, NULL)) return 1;
        return spit(argv[4], out, m) ? 1 : 0;
    }
    if (argc == 4 && !strcmp(argv[1], "decode")) {
        size_t n, m;
        uint8_t *in = slurp(argv[2], &n), *out;
        if (!in || orc_decode_mem(in, n, &out, &m, NULL)) { fprintf(stderr, "decoding failed\n"); return 1; }
        return spit(argv[3], out, m) ? 1 : 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "time") && !level_cfg(argv[2], &cfg)) {
        size_t n, m = 0;
        uint8_t *in = slurp(argv[3], &n), *out;
        int reps = argc > 4 ? atoi(argv[4]) : 1;
        double best = 1e30;
        if (!in) return 1;
        for (int r = 0; r < reps; r++) {
            double t0 = now();
            if (orc_encode_mem(in, n, &cfg, &out, &m, NULL)) return 1;
            double t = now() - t0;
            if (t < best) best = t;
            orc_free(out);
        }
        printf("{\"in_bytes\": %zu, \"out_bytes\": %zu, \"seconds\": %.6f, \"mb_per_s\": %.3f}\n", n, m, best,
               n / best / 1e6);
        return 0;
    }
    fprintf(stderr, "usage: orz_oracle encode -lN IN OUT | decode IN OUT | time -lN IN [reps]\n");
    return 2;
}

// dev tool: times orz_symrank_kernel on a recorded item stream and checks the ranks (tools/dev/make_symrank_case.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../orz_amd/csrc/backend_hip.h"
#include "symrank_r3.h"
using namespace orz;
int main(int argc, char** argv) {
    FILE* f = fopen(argc > 1 ? argv[1] : "build/symrank_case.bin", "rb");
    if (!f) { perror("case"); return 1; }
    uint32_t n; fread(&n, 4, 1, f);
    std::vector<uint32_t> gsym(n), rstart(513); std::vector<uint16_t> state(512 * kSrWords), want(n), got(n);
    fread(gsym.data(), 4, n, f); fread(rstart.data(), 4, 513, f); fread(state.data(), 2, state.size(), f); fread(want.data(), 2, n, f);
    // optional: replace every context's items by a cycle over the N most frequent symbols (path timing: N <= 40 stays in ranks 0..63, ~100 in 64..127, ...)
    const int cyc = argc > 3 && !strcmp(argv[2], "cyc") ? atoi(argv[3]) : 0;
    if (cyc) for (int c = 0; c < 512; c++) for (uint32_t j = rstart[c]; j < rstart[c + 1]; j++) gsym[j] = state[(size_t)c * kSrWords + (j - rstart[c]) % cyc] | (388u << 16);
    uint32_t *dg, *dr; uint16_t *ds, *dk;
    hipMalloc(&dg, n * 4); hipMalloc(&dr, 513 * 4); hipMalloc(&ds, state.size() * 2); hipMalloc(&dk, n * 2);
    hipMemcpy(dg, gsym.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dr, rstart.data(), 513 * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    uint32_t hot = 0; for (int c = 0; c < 512; c++) if (rstart[c + 1] - rstart[c] > hot) hot = rstart[c + 1] - rstart[c];
    int rc = 0;
    std::vector<uint16_t> st_old(state.size()), st_new(state.size());
    for (int which = 0; which < 2; which++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipMemcpy(ds, state.data(), state.size() * 2, hipMemcpyHostToDevice);
            hipMemset(dk, 0xee, n * 2);
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(orz_symrank_kernel_r3, dim3(512), dim3(64), 0, 0, ds, dg, dk, dr, (const uint16_t*)nullptr, (const uint32_t*)nullptr);
            else hipLaunchKernelGGL(orz_symrank_kernel, dim3(512), dim3(64), 0, 0, ds, dg, dk, dr, (const uint16_t*)nullptr, (const uint32_t*)nullptr);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        hipMemcpy(got.data(), dk, n * 2, hipMemcpyDeviceToHost);
        hipMemcpy((which ? st_new : st_old).data(), ds, state.size() * 2, hipMemcpyDeviceToHost);
        size_t bad = 0, first = n; for (uint32_t i = 0; i < n; i++) if (got[i] != want[i]) { if (!bad) first = i; bad++; }
        printf("%s: items %u hottest %u  kernel %.2f ms  = %.1f ns per item of the hottest context  mismatches %zu (first at %zu)\n",
               which ? "lanes" : "r3   ", n, hot, best, best * 1e6 / hot, bad, first);
        if (bad && !cyc) {
            rc = 1;
            int c = 0; while (rstart[c + 1] <= first) c++;
            printf("   first mismatch: context %d, item %zu of %u in it: got %u want %u (sym %u unl %u)\n", c, first - rstart[c], rstart[c + 1] - rstart[c], got[first], want[first], gsym[first] & 0xffff, gsym[first] >> 16);
        }
    }
    size_t sd = 0; for (size_t i = 0; i < state.size(); i++) sd += st_old[i] != st_new[i];
    printf("final tables: %zu words differ between the two kernels\n", sd);
    return rc || sd != 0;
}

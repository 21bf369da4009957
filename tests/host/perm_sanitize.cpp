// Host-only harness: the permutation pipeline under ThreadSanitizer / AddressSanitizer.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ppo2hip.h"
namespace ppo2hip { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }

int main() {
    const int64_t n = 20000;
    const int E = 4;
    ppo2hip_rng *a = ppo2hip_rng_create(7), *b = ppo2hip_rng_create(7);
    ppo2hip_perm *p = ppo2hip_perm_create(a, n, E);
    std::vector<std::vector<int32_t>> bufs(E, std::vector<int32_t>(n));
    std::vector<int32_t *> ptrs;
    for (auto &v : bufs) ptrs.push_back(v.data());
    std::vector<int32_t> ref(n), cur(n), tmp(n);
    int bad = 0;
    for (int job = 0; job < 40; job++) {
        const bool rel = job % 2;
        if ((rel ? ppo2hip_perm_begin_relative : ppo2hip_perm_begin)(p, E, ptrs.data())) return 2;
        for (int64_t i = 0; i < n; i++) ref[i] = (int32_t)i;
        for (int e = 0; e < E; e++) {
            if (ppo2hip_perm_wait(p, e)) return 3;
            ppo2hip_shuffle_i32(b, ref.data(), n);
            if (!rel) {
                bad += memcmp(bufs[e].data(), ref.data(), n * 4) != 0;
            } else {
                if (e == 0) cur = bufs[0];
                else { for (int64_t i = 0; i < n; i++) tmp[i] = cur[bufs[e][i]]; cur.swap(tmp); }
                bad += memcmp(cur.data(), ref.data(), n * 4) != 0;
            }
        }
        if (ppo2hip_perm_wait_all(p)) return 4;
    }
    ppo2hip_perm_destroy(p);
    ppo2hip_rng_destroy(a);
    ppo2hip_rng_destroy(b);
    printf("mismatching epochs: %d\n", bad);
    return bad != 0;
}

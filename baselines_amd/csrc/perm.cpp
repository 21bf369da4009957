// Host-side minibatch permutation generator, bit-exact with NumPy's legacy RandomState:
// MT19937 (init_genrand seeding) and the Fisher-Yates pass of RandomState.shuffle whose
// bounded draws mask a 32-bit output to the next power of two minus one and reject values
// above the bound.  This is what np.random.seed / np.random.shuffle do at
// baselines/ppo2/ppo2.py:113 and :130 (seeded through baselines/common/misc_util.py:69-76).
//
// The pass is split in two stages that touch disjoint state, so they pipeline without
// changing the stream or the result:
//   draw  : MT19937 outputs (generated 624 at a time, tempering vectorised) -> accepted swap
//           targets j_i, compacted without a data-dependent branch (the accept test only
//           steers an index increment);
//   swap  : x[i] <-> x[j_i] with the target cache lines prefetched.
// ppo2hip_shuffle_* runs both stages inline; the ppo2hip_perm_* pipeline runs them on worker
// threads and produces the cumulative shuffles of all epochs of an update ahead of the GPU,
// so index generation overlaps the kernels of the previous epoch.
#include <algorithm>
#include <chrono>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define PPO2HIP_X86 1
#endif

#include "common.hpp"

struct ppo2hip_rng {
    uint32_t mt[624];
    int pos;           // NumPy's position: next output is tempered mt[pos]; 624 = regenerate first
};

namespace {

constexpr int N = 624, M = 397;

void seed_state(ppo2hip_rng *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < N; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = N;
}

inline uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
}

inline uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// Next 624-word block of the generator, and its tempered outputs.  The body is compiled twice on
// x86: for the baseline vector width and for 512-bit vectors (picked at run time).
//   words 0..226 read only not-yet-rewritten words: independent iterations;
//   words 227..622 read words rewritten 227 iterations earlier: safe for any vector width <= 227.
#define PPO2HIP_MT_BODY                                                                          \
    if (advance) {                                                                               \
        int k = 0;                                                                               \
        _Pragma("clang loop vectorize(assume_safety)")                                           \
        for (k = 0; k < N - M; k++) mt[k] = mt[k + M] ^ twist(mt[k], mt[k + 1]);                 \
        _Pragma("clang loop vectorize(assume_safety)")                                           \
        for (k = N - M; k < N - 1; k++) mt[k] = mt[k + (M - N)] ^ twist(mt[k], mt[k + 1]);       \
        mt[N - 1] = mt[M - 1] ^ twist(mt[N - 1], mt[0]);                                         \
    }                                                                                            \
    if (out)                                                                                     \
        for (int i = 0; i < N; i++) out[i] = temper(mt[i]);

void next_block_base(uint32_t *__restrict mt, uint32_t *__restrict out, bool advance) { PPO2HIP_MT_BODY }
#if defined(PPO2HIP_X86)
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq")))
void next_block_512(uint32_t *__restrict mt, uint32_t *__restrict out, bool advance) { PPO2HIP_MT_BODY }
bool cpu_has_avx512() {
    // PPO2HIP_NO_AVX512 (any value) keeps the baseline path, so that both can be tested on one machine
    static const bool yes = !getenv("PPO2HIP_NO_AVX512") && __builtin_cpu_supports("avx512f") &&
                            __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") &&
                            __builtin_cpu_supports("avx512dq");
    return yes;
}
#else
bool cpu_has_avx512() { return false; }
#endif

inline void next_block(uint32_t *mt, uint32_t *out, bool advance) {
#if defined(PPO2HIP_X86)
    if (cpu_has_avx512()) return next_block_512(mt, out, advance);
#endif
    next_block_base(mt, out, advance);
}

void regenerate(uint32_t *mt) { next_block(mt, nullptr, true); }

// Buffered view of the stream, one 624-word block of tempered outputs at a time: out[pos ..)
// are the unread outputs of the current block, refill() moves to the next block.
struct InlineStream {
    ppo2hip_rng *s;
    uint32_t buf[N];
    const uint32_t *out = buf;
    int pos;
    explicit InlineStream(ppo2hip_rng *r) : s(r), pos(r->pos) {
        if (pos < N) fill();
    }
    ~InlineStream() { s->pos = pos; }
    void fill() { next_block(s->mt, buf, false); }
    void refill() {
        next_block(s->mt, buf, true);
        pos = 0;
    }
};

#if defined(PPO2HIP_X86) && defined(__AVX2__)
// lane permutation that packs the lanes selected by an 8-bit mask to the front
struct PackLut {
    alignas(32) uint32_t idx[256][8];
    PackLut() {
        for (int m = 0; m < 256; m++) {
            int w = 0;
            for (int b = 0; b < 8; b++)
                if (m & (1 << b)) idx[m][w++] = b;
            for (; w < 8; w++) idx[m][w] = 0;
        }
    }
};
static const PackLut g_pack;
#endif

// Accepted swap targets for positions hi, hi-1, ..., lo (lo >= 1), written to js[0 .. hi-lo]
// (up to 16 words of scratch may be written past that).  Eight candidates are classified at
// once: a masked value <= i-7 is accepted and one > i is rejected whatever the seven
// decisions before it were, so unless a value falls in the 7-wide window between (about
// 8 / 2^k of the blocks) the block is compacted with one vector permute; otherwise, and
// near buffer or mask-range ends, candidates go through the scalar loop, in which the
// accept test only steers an index increment (no data-dependent branch).
#if defined(PPO2HIP_X86)
// 64 candidates at a time on 512-bit vectors; same decisions as the 256-bit loop in draw_targets
// (a value <= i-63 is accepted, one > i rejected, anything between ends the run).  Returns the
// advanced (p, i, w) through the references.
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq")))
void draw_run_512(const uint32_t *out, int &p, int64_t &i, int64_t stop, uint32_t mask, uint32_t *js, int64_t &w) {
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    while (p + 64 <= N && i - 63 >= stop) {
        const __m512i t_lo = _mm512_set1_epi32((int)(i - 63)), t_hi = _mm512_set1_epi32((int)i);
        __m512i v[4];
        __mmask16 acc[4], amb = 0;
        for (int b = 0; b < 4; b++) {
            v[b] = _mm512_and_si512(_mm512_loadu_si512((const void *)(out + p + 16 * b)), vmask);
            const __mmask16 gt_lo = _mm512_cmpgt_epi32_mask(v[b], t_lo);
            amb |= gt_lo ^ _mm512_cmpgt_epi32_mask(v[b], t_hi);
            acc[b] = (__mmask16)~gt_lo;
        }
        if (amb) break;
        const int64_t w0 = w;
        for (int b = 0; b < 4; b++) {
            // compress in a register, then a plain store (the memory form of the compress is slow on some cores)
            _mm512_storeu_si512((void *)(js + w), _mm512_maskz_compress_epi32(acc[b], v[b]));
            w += __builtin_popcount((unsigned)acc[b]);
        }
        i -= w - w0;
        p += 64;
    }
}
#endif

template <class S>
void draw_targets(S &st, int64_t hi, int64_t lo, uint32_t *js) {
    int64_t i = hi;
    int64_t w = 0;
    while (i >= lo) {
        if (st.pos >= N) st.refill();
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        // positions that share this mask: i down to max(lo, (mask >> 1) + 1)
        const int64_t lim_mask = (int64_t)(mask >> 1) + 1;
        const int64_t stop = lim_mask > lo ? lim_mask : lo;
        int p = st.pos;
#if defined(PPO2HIP_X86)
        if (mask <= 0x7fffffffu && cpu_has_avx512()) draw_run_512(st.out, p, i, stop, mask, js, w);
#endif
#if defined(PPO2HIP_X86) && defined(__AVX2__)
        if (mask <= 0x7fffffffu) {
            const __m256i vmask = _mm256_set1_epi32((int)mask);
            // 64 candidates against thresholds fixed on entry: the k-th one's true bound lies in
            // [i-63, i], so a value <= i-63 is accepted and one > i rejected whatever happens
            // before it; with no value in between (all but ~64 * 63 / 2^k of the groups) the
            // eight blocks are classified with no dependence on each other's accept counts.
            while (p + 64 <= N && i - 63 >= stop) {
                const __m256i t_lo = _mm256_set1_epi32((int)(i - 63)), t_hi = _mm256_set1_epi32((int)i);
                __m256i v[8], amb = _mm256_setzero_si256();
                int acc[8];
                for (int b = 0; b < 8; b++) {
                    v[b] = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(st.out + p + 8 * b)), vmask);
                    const __m256i gt_lo = _mm256_cmpgt_epi32(v[b], t_lo);
                    amb = _mm256_or_si256(amb, _mm256_xor_si256(gt_lo, _mm256_cmpgt_epi32(v[b], t_hi)));
                    acc[b] = (~_mm256_movemask_ps(_mm256_castsi256_ps(gt_lo))) & 0xff;
                }
                if (!_mm256_testz_si256(amb, amb)) break;     // decide this stretch block by block
                const int64_t w0 = w;
                for (int b = 0; b < 8; b++) {
                    _mm256_storeu_si256((__m256i *)(js + w),
                                        _mm256_permutevar8x32_epi32(v[b], _mm256_load_si256((const __m256i *)g_pack.idx[acc[b]])));
                    w += __builtin_popcount(acc[b]);
                }
                i -= w - w0;
                p += 64;
            }
            while (p + 8 <= N && i - 7 >= stop) {
                const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(st.out + p)), vmask);
                const __m256i gt_lo = _mm256_cmpgt_epi32(v, _mm256_set1_epi32((int)(i - 7)));   // v > i-7
                const __m256i gt_hi = _mm256_cmpgt_epi32(v, _mm256_set1_epi32((int)i));         // v > i
                const int m_gt_lo = _mm256_movemask_ps(_mm256_castsi256_ps(gt_lo));
                const int m_gt_hi = _mm256_movemask_ps(_mm256_castsi256_ps(gt_hi));
                if (m_gt_lo != m_gt_hi) break;            // some value in (i-7, i]: decide serially
                const int acc = (~m_gt_lo) & 0xff;
                const __m256i packed = _mm256_permutevar8x32_epi32(v, _mm256_load_si256((const __m256i *)g_pack.idx[acc]));
                _mm256_storeu_si256((__m256i *)(js + w), packed);
                const int na = __builtin_popcount(acc);
                w += na;
                i -= na;
                p += 8;
            }
        }
#endif
        // scalar: at most 8 candidates after a vector bail-out, everything near the ends
        int budget = 8;
        while (p < N && i >= stop && budget-- > 0) {
            const uint32_t v = st.out[p++] & mask;
            js[w] = v;
            const int64_t acc = (int64_t)(v <= (uint32_t)i);
            w += acc;
            i -= acc;
        }
        st.pos = p;
    }
}

template <class I>
void apply_swaps(I *x, int64_t hi, const uint32_t *js, int64_t cnt) {
    constexpr int AHEAD = 24;
    for (int64_t k = 0; k < cnt; k++) {
        if (k + AHEAD < cnt) __builtin_prefetch(&x[js[k + AHEAD]], 1, 1);
        const int64_t a = hi - k;
        const I tmp = x[a];
        x[a] = x[js[k]];
        x[js[k]] = tmp;
    }
}

template <class I>
int shuffle_impl(ppo2hip_rng *rng, I *x, int64_t n) {
    if (!rng || (!x && n > 0)) {
        ppo2hip::set_error("shuffle: null pointer");
        return PPO2HIP_EINVAL;
    }
    if (n < 0 || n > 0xffffffffll) {
        ppo2hip::set_error("shuffle: n=%lld outside [0, 2^32)", (long long)n);
        return PPO2HIP_EINVAL;
    }
    constexpr int64_t BLK = 4096;
    std::vector<uint32_t> js(BLK + 16);
    InlineStream st(rng);
    int64_t i = n - 1;
    while (i >= 1) {
        const int64_t cnt = i < BLK ? i : BLK;
        draw_targets(st, i, i - cnt + 1, js.data());
        apply_swaps(x, i, js.data(), cnt);
        i -= cnt;
    }
    return PPO2HIP_OK;
}

}  // namespace

extern "C" ppo2hip_rng *ppo2hip_rng_create(uint32_t seed) {
    ppo2hip_rng *r = new ppo2hip_rng;
    seed_state(r, seed);
    return r;
}
extern "C" void ppo2hip_rng_destroy(ppo2hip_rng *rng) { delete rng; }
extern "C" void ppo2hip_rng_seed(ppo2hip_rng *rng, uint32_t seed) {
    if (rng) seed_state(rng, seed);
}
extern "C" int ppo2hip_rng_set_state(ppo2hip_rng *rng, const uint32_t *key, int pos) {
    if (!rng || !key || pos < 0 || pos > N) {
        ppo2hip::set_error("rng_set_state: bad argument (pos=%d)", pos);
        return PPO2HIP_EINVAL;
    }
    std::memcpy(rng->mt, key, sizeof(rng->mt));
    rng->pos = pos;
    return PPO2HIP_OK;
}
extern "C" int ppo2hip_rng_get_state(const ppo2hip_rng *rng, uint32_t *key, int *pos) {
    if (!rng || !key || !pos) {
        ppo2hip::set_error("rng_get_state: null pointer");
        return PPO2HIP_EINVAL;
    }
    std::memcpy(key, rng->mt, sizeof(rng->mt));
    *pos = rng->pos;
    return PPO2HIP_OK;
}
extern "C" int ppo2hip_shuffle_i64(ppo2hip_rng *rng, int64_t *inds, int64_t n) { return shuffle_impl(rng, inds, n); }
extern "C" int ppo2hip_shuffle_i32(ppo2hip_rng *rng, int32_t *inds, int64_t n) { return shuffle_impl(rng, inds, n); }

// ---------------------------------------------------------------------------------------
// Pipeline over all epochs of an update
//
// Shuffling an array x in place leaves x o sigma, where sigma is what the same swaps make of
// the identity.  So the epochs' Fisher-Yates passes, which the reference chains through one
// array (epoch e shuffles epoch e-1's order), split into per-epoch jobs that workers take in
// turn (epoch e goes to worker e mod W):
//   draw    : the worker whose turn it is owns the generator and draws its epoch's swap targets,
//             in stream order, into a buffer of its own, then passes the generator on — the only
//             data that moves between cores is the 2.5 KB generator state;
//   swap    : sigma_e from the identity with those targets, while the next worker draws;
//   compose : out_e = out_{e-1}[sigma_e] (out_0 = sigma_0), a gather, chained in epoch order; in
//             "relative" mode the sigma_e themselves are handed out and the caller composes
//             (ppo2hip_compose_i32 does it on the device).
// ---------------------------------------------------------------------------------------
struct ppo2hip_perm {
    static constexpr int kMaxWorkers = 4;
    int spin_job_us = 0, spin_stage_us = 2000;   // how long a wait polls before it blocks (TEMP: env-tunable)
    ppo2hip_rng *rng = nullptr;
    int64_t n = 0;
    int max_epochs = 0;
    int nworkers = 1;
    std::vector<uint32_t> js[kMaxWorkers];     // the worker's swap targets in draw order, [n-1]
    std::vector<int32_t> sigma[kMaxWorkers];
    std::vector<int32_t *> out;            // caller's per-epoch destinations
    int nepochs = 0;
    bool relative = false;                 // out[e] = sigma_e (the caller composes) instead of the running order
    // progress flags: written under mu (so that blocked waiters can be notified), read lock-free by pollers
    std::atomic<uint64_t> job{0};
    std::atomic<int> epochs_drawn{0};      // epochs drawn so far = the epoch whose worker owns the generator
    std::unique_ptr<std::atomic<char>[]> epoch_done;   // out[e] complete (epochs finish in any order)
    std::atomic<int> workers_idle{0};      // workers that finished the current job
    std::atomic<bool> quit{false};
    std::mutex mu;
    std::condition_variable cv;
    std::thread t_work[kMaxWorkers];

    // Wait for pred (a function of the flags above).  The stages of a job follow each other within
    // a fraction of a millisecond, and a thread that blocks gives its CPU away and restarts cold (on
    // a virtual machine that alone costs as much as a stage), so poll first and block only when
    // nothing comes for a while, e.g. between updates of a slow environment.
    template <class Pred>
    void await(Pred pred, int kSpinMicros) {
        const auto t0 = std::chrono::steady_clock::now();
        if (kSpinMicros > 0)
        for (int k = 1;; k++) {
            if (pred()) return;
#if defined(PPO2HIP_X86)
            for (int q = 0; q < 32; q++) _mm_pause();
#endif
            if ((k & 31) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinMicros)) break;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, pred);
    }
    template <class F>
    void publish(F set) {
        {
            std::lock_guard<std::mutex> lk(mu);
            set();
        }
        cv.notify_all();
    }

    void draw_epoch(uint32_t *dst) {
        InlineStream st(rng);              // hands its position back to the generator when it goes out of scope
        constexpr int64_t BLK = 8192;
        const int64_t per = n - 1;
        int64_t i = n - 1;
        while (i >= 1) {
            const int64_t cnt = i < BLK ? i : BLK;
            draw_targets(st, i, i - cnt + 1, dst + (per - i));
            i -= cnt;
        }
    }

    void work_loop(int w) {
        uint64_t seen = 0;
        for (;;) {
            await([&] { return quit.load() || job.load(std::memory_order_acquire) != seen; }, spin_job_us);
            if (quit.load()) return;
            seen = job.load(std::memory_order_acquire);
            for (int e = w; e < nepochs; e += nworkers) {
                int32_t *sg = (relative || e == 0) ? out[e] : sigma[w].data();
                for (int64_t k = 0; k < n; k++) sg[k] = (int32_t)k;
                await([&] { return epochs_drawn.load(std::memory_order_acquire) == e; }, spin_stage_us);    // the generator is ours
                draw_epoch(js[w].data());
                publish([&] { epochs_drawn.store(e + 1, std::memory_order_release); });      // ... and the next worker's
                apply_swaps(sg, n - 1, js[w].data(), n - 1);
                if (!relative && e > 0) {
                    await([&] { return epoch_done[e - 1].load(std::memory_order_acquire) != 0; }, spin_stage_us);
                    const int32_t *prev = out[e - 1];
                    int32_t *dst = out[e];
                    constexpr int AHEAD = 32;
                    for (int64_t k = 0; k < n; k++) {
                        if (k + AHEAD < n) __builtin_prefetch(&prev[sg[k + AHEAD]], 0, 1);
                        dst[k] = prev[sg[k]];
                    }
                }
                publish([&] { epoch_done[e].store(1, std::memory_order_release); });
            }
            publish([&] { workers_idle.fetch_add(1, std::memory_order_release); });
        }
    }
};

extern "C" ppo2hip_perm *ppo2hip_perm_create(ppo2hip_rng *rng, int64_t n, int max_epochs) {
    if (!rng || n < 1 || n > 0x7fffffffll || max_epochs < 1 || max_epochs > 1024) {
        ppo2hip::set_error("perm_create: bad argument (n=%lld, max_epochs=%d)", (long long)n, max_epochs);
        return nullptr;
    }
    ppo2hip_perm *p = new ppo2hip_perm;
    p->rng = rng;
    p->n = n;
    p->max_epochs = max_epochs;
    p->nworkers = max_epochs < ppo2hip_perm::kMaxWorkers ? max_epochs : ppo2hip_perm::kMaxWorkers;
    p->workers_idle.store(p->nworkers);
    if (getenv("PPO2HIP_SPIN_JOB_US")) p->spin_job_us = atoi(getenv("PPO2HIP_SPIN_JOB_US"));
    if (getenv("PPO2HIP_SPIN_STAGE_US")) p->spin_stage_us = atoi(getenv("PPO2HIP_SPIN_STAGE_US"));
    p->out.resize(max_epochs);
    p->epoch_done.reset(new std::atomic<char>[max_epochs]);
    for (int e = 0; e < max_epochs; e++) p->epoch_done[e].store(0);
    for (int w = 0; w < p->nworkers; w++) {
        p->js[w].resize((size_t)(n > 1 ? n - 1 : 1) + 16);
        p->sigma[w].resize((size_t)n);
        p->t_work[w] = std::thread([p, w] { p->work_loop(w); });
    }
    return p;
}

extern "C" int ppo2hip_perm_wait_all(ppo2hip_perm *p);

extern "C" void ppo2hip_perm_destroy(ppo2hip_perm *p) {
    if (!p) return;
    ppo2hip_perm_wait_all(p);
    p->publish([&] { p->quit.store(true); });
    for (int w = 0; w < p->nworkers; w++) p->t_work[w].join();
    delete p;
}

static int perm_begin(ppo2hip_perm *p, int nepochs, int32_t *const *out, bool relative) {
    if (!p || !out || nepochs < 1 || nepochs > p->max_epochs) {
        ppo2hip::set_error("perm_begin: bad argument");
        return PPO2HIP_EINVAL;
    }
    for (int e = 0; e < nepochs; e++)
        if (!out[e]) {
            ppo2hip::set_error("perm_begin: null destination for epoch %d", e);
            return PPO2HIP_EINVAL;
        }
    int rc = ppo2hip_perm_wait_all(p);   // one job at a time: the previous one must have drained
    if (rc) return rc;
    p->publish([&] {
        p->nepochs = nepochs;
        p->relative = relative;
        for (int e = 0; e < nepochs; e++) p->out[e] = out[e];
        p->epochs_drawn.store(0);
        for (int e = 0; e < p->max_epochs; e++) p->epoch_done[e].store(0);
        p->workers_idle.store(0);
        p->job.fetch_add(1, std::memory_order_release);
    });
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_perm_begin(ppo2hip_perm *p, int nepochs, int32_t *const *out) {
    return perm_begin(p, nepochs, out, false);
}
extern "C" int ppo2hip_perm_begin_relative(ppo2hip_perm *p, int nepochs, int32_t *const *out) {
    return perm_begin(p, nepochs, out, true);
}

extern "C" int ppo2hip_perm_wait(ppo2hip_perm *p, int epoch) {
    if (!p || epoch < 0 || epoch >= p->nepochs) {
        ppo2hip::set_error("perm_wait: bad epoch");
        return PPO2HIP_EINVAL;
    }
    p->await([&] { return p->epoch_done[epoch].load(std::memory_order_acquire) != 0; }, p->spin_stage_us);
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_perm_wait_drawn(ppo2hip_perm *p) {
    if (!p) {
        ppo2hip::set_error("perm_wait_drawn: null handle");
        return PPO2HIP_EINVAL;
    }
    // every epoch of the job has drawn its swap targets: the generator is free, although swaps may still be running
    p->await([&] { return p->nepochs == 0 || p->epochs_drawn.load(std::memory_order_acquire) >= p->nepochs; }, p->spin_stage_us);
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_perm_wait_all(ppo2hip_perm *p) {
    if (!p) {
        ppo2hip::set_error("perm_wait_all: null handle");
        return PPO2HIP_EINVAL;
    }
    p->await([&] { return p->workers_idle.load(std::memory_order_acquire) == p->nworkers; }, p->spin_stage_us);
    return PPO2HIP_OK;
}

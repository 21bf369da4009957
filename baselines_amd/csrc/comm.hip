// Data-parallel exchange behind the C ABI: the role of MpiAdam's MPI calls (baselines/common/mpi_adam.py:21-50)
// with RCCL over xGMI on device buffers.  One communicator per process (= per GPU); every call is queued on the
// caller's stream, so a whole epoch of gradient -> all-reduce -> Adam steps runs without the host in the loop.
//
// RCCL is bound at run time (dlopen), first to the copy already loaded into the process (a host framework that
// bundles its own build must not end up with two RCCLs talking to one HIP runtime), then by soname.  A job of one
// rank needs no RCCL at all; passing a unique id asks for RCCL even then (a one-rank communicator), which is how the
// binding is exercised on a one-GPU box.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "kernels.hpp"

struct ppo2hip_comm {
    int rank, size;
    ncclComm_t nccl;      // null: single rank without RCCL
};

namespace ppo2hip {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static bool rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return true;
    const char *loaded[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : loaded)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already holds
    const char *fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : fresh)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so not found: %s", dlerror());
        return false;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) { ok = false; snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so lacks %s", name); }
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) return false;
    g_rccl.handle = h;
    return true;
}

#define PPO2HIP_CHECK_NCCL(expr)                                                                      \
    do {                                                                                              \
        ncclResult_t r__ = (expr);                                                                    \
        if (r__ != ncclSuccess) {                                                                     \
            set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); \
            return PPO2HIP_ECOMM;                                                                     \
        }                                                                                             \
    } while (0)

// Order-independent 64-bit digests of the parameter bits: h1 = sum of the words, h2 = sum of word * (odd multiplier
// of the index).  out = {h1, h2, ~h1, ~h2}: after a MAX all-reduce the replicas agree iff max(h) == ~max(~h) = min(h).
__global__ __launch_bounds__(1024) void digest_kernel(const uint32_t *__restrict__ w, int64_t n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s1[16], s2[16];
    unsigned long long h1 = 0, h2 = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long x = w[i];
        h1 += x;
        h2 += x * (unsigned long long)(2654435761u * (uint32_t)i + 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        h1 += __shfl_xor(h1, o, 64);
        h2 += __shfl_xor(h2, o, 64);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s1[wv] = h1; s2[wv] = h2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) { a += s1[i]; b += s2[i]; }
        out[0] = a; out[1] = b; out[2] = ~a; out[3] = ~b;
    }
}
__global__ void digest_verdict_kernel(const unsigned long long *__restrict__ d, int32_t *__restrict__ synced) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *synced = (d[0] == ~d[2] && d[1] == ~d[3]) ? 1 : 0;
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" int ppo2hip_comm_unique_id(void *id_out) {
    PPO2HIP_REQUIRE(id_out, "comm_unique_id: null pointer");
    if (!rccl_load()) { set_error("comm_unique_id: %s", g_rccl.why); return PPO2HIP_ECOMM; }
    static_assert(sizeof(ncclUniqueId) == PPO2HIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    PPO2HIP_CHECK_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return PPO2HIP_OK;
}

extern "C" ppo2hip_comm *ppo2hip_comm_create(const void *unique_id, int rank, int size) {
    if (size < 1 || rank < 0 || rank >= size) { set_error("comm_create: rank %d of %d", rank, size); return nullptr; }
    if (!unique_id && size > 1) { set_error("comm_create: a job of %d ranks needs rank 0's unique id", size); return nullptr; }
    ppo2hip_comm *c = new ppo2hip_comm{rank, size, nullptr};
    if (!unique_id) return c;
    if (!rccl_load()) { set_error("comm_create: %s", g_rccl.why); delete c; return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, size, id, rank);      // on the calling thread's current device
    if (r != ncclSuccess) {
        set_error("comm_create: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void ppo2hip_comm_destroy(ppo2hip_comm *c) {
    if (!c) return;
    if (c->nccl) g_rccl.CommDestroy(c->nccl);
    delete c;
}

extern "C" int ppo2hip_comm_rank(const ppo2hip_comm *c) { return c ? c->rank : 0; }
extern "C" int ppo2hip_comm_size(const ppo2hip_comm *c) { return c ? c->size : 1; }

static int allreduce(ppo2hip_comm *c, void *buf, int64_t n, ncclDataType_t dt, ncclRedOp_t op, void *stream, const char *what) {
    PPO2HIP_REQUIRE(n >= 0, "%s: negative count", what);
    if (!c || !c->nccl || n == 0) return PPO2HIP_OK;        // one rank: the sum over ranks is the buffer itself
    PPO2HIP_REQUIRE(buf, "%s: null buffer", what);
    PPO2HIP_CHECK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, dt, op, c->nccl, (hipStream_t)stream));
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_allreduce_grads(ppo2hip_comm *c, float *buf, int64_t n, void *stream) {
    return allreduce(c, buf, n, ncclFloat32, ncclSum, stream, "allreduce_grads");
}

extern "C" int ppo2hip_allreduce_f64(ppo2hip_comm *c, double *buf, int64_t n, void *stream) {
    return allreduce(c, buf, n, ncclFloat64, ncclSum, stream, "allreduce_f64");
}

extern "C" int ppo2hip_bcast_params(ppo2hip_comm *c, float *buf, int64_t n, int root, void *stream) {
    PPO2HIP_REQUIRE(n >= 0 && root >= 0 && root < ppo2hip_comm_size(c), "bcast_params: bad count or root");
    if (!c || !c->nccl || n == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(buf, "bcast_params: null buffer");
    PPO2HIP_CHECK_NCCL(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->nccl, (hipStream_t)stream));
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_replica_checksum(ppo2hip_comm *c, const float *params, int64_t n, void *workspace, int32_t *synced,
                                        void *stream) {
    PPO2HIP_REQUIRE(params && workspace && synced && n >= 1, "replica_checksum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d = (unsigned long long *)workspace;
    hipLaunchKernelGGL(digest_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t *)params, n, d);
    int rc = check_launch("digest_kernel");
    if (rc) return rc;
    if (c && c->nccl) PPO2HIP_CHECK_NCCL(g_rccl.AllReduce(d, d, 4, ncclUint64, ncclMax, c->nccl, s));
    hipLaunchKernelGGL(digest_verdict_kernel, dim3(1), dim3(64), 0, s, d, synced);
    return check_launch("digest_verdict_kernel");
}

extern "C" int ppo2hip_ppo_epoch_dp(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, float *params,
                                    float *grads, float *adam_m, float *adam_v, const ppo2hip_rollout *rollout,
                                    const int32_t *idx, int64_t N, int64_t nbatch_train, const double *stats,
                                    float cliprange, double lr, int64_t t_first, double beta1, double beta2,
                                    double epsilon, double max_grad_norm, float *losses, void *workspace,
                                    size_t workspace_bytes, void *adam_scratch, ppo2hip_comm *comm, double grad_scale,
                                    void *stream) {
    PPO2HIP_REQUIRE(desc && idx && adam_m && adam_v && adam_scratch && losses, "ppo_epoch_dp: null pointer");
    PPO2HIP_REQUIRE(nbatch_train >= 1 && N >= nbatch_train && N % nbatch_train == 0,
                    "ppo_epoch_dp: N=%lld is not a multiple of nbatch_train=%lld", (long long)N, (long long)nbatch_train);
    PPO2HIP_REQUIRE(desc->kind == PPO2HIP_POLICY_MLP || desc->kind == PPO2HIP_POLICY_CNN,
                    "ppo_epoch_dp: feed-forward policies only (recurrent minibatches are env sequences)");
    const int64_t nmb = N / nbatch_train;
    const int64_t P = ppo2hip_param_count(desc);
    if (P < 0) return PPO2HIP_EINVAL;
    for (int64_t j = 0; j < nmb; j++) {
        int rc = ppo2hip_ppo_grad(desc, loss, params, rollout, idx + j * nbatch_train, nbatch_train, nullptr, 0,
                                  stats ? stats + 3 * j : nullptr, cliprange, 1.0 / (double)nbatch_train, grads,
                                  losses + 8 * j, workspace, workspace_bytes, stream);
        if (rc) return rc;
        if ((rc = ppo2hip_allreduce_grads(comm, grads, P, stream))) return rc;
        rc = ppo2hip_adam_step(params, grads, adam_m, adam_v, P, lr, t_first + j, beta1, beta2, epsilon, grad_scale,
                               max_grad_norm, adam_scratch, stream);
        if (rc) return rc;
    }
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_ppo_epoch_shares(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, float *params,
                                        float *grads, float *adam_m, float *adam_v, const ppo2hip_rollout *rollout,
                                        const int32_t *idx, int64_t idx_stride, const int64_t *counts, int nmb,
                                        const double *stats, float cliprange, double inv_count, double lr,
                                        int64_t t_first, double beta1, double beta2, double epsilon,
                                        double max_grad_norm, float *losses, void *workspace, size_t workspace_bytes,
                                        void *adam_scratch, ppo2hip_comm *comm, void *stream) {
    PPO2HIP_REQUIRE(desc && idx && counts && adam_m && adam_v && adam_scratch && losses && grads, "ppo_epoch_shares: null pointer");
    PPO2HIP_REQUIRE(nmb >= 1 && idx_stride >= 1, "ppo_epoch_shares: nmb=%d idx_stride=%lld", nmb, (long long)idx_stride);
    PPO2HIP_REQUIRE(desc->kind == PPO2HIP_POLICY_MLP || desc->kind == PPO2HIP_POLICY_CNN,
                    "ppo_epoch_shares: feed-forward policies only (recurrent minibatches are env sequences)");
    const int64_t P = ppo2hip_param_count(desc);
    if (P < 0) return PPO2HIP_EINVAL;
    for (int j = 0; j < nmb; j++) {
        PPO2HIP_REQUIRE(counts[j] >= 0 && counts[j] <= idx_stride, "ppo_epoch_shares: count %lld of minibatch %d outside [0, %lld]",
                        (long long)counts[j], j, (long long)idx_stride);
        int rc;
        if (counts[j] > 0) {
            rc = ppo2hip_ppo_grad(desc, loss, params, rollout, idx + (int64_t)j * idx_stride, counts[j], nullptr, 0,
                                  stats ? stats + 3 * j : nullptr, cliprange, inv_count, grads, losses + 8 * j, workspace,
                                  workspace_bytes, stream);
            if (rc) return rc;
        } else {        // none of this minibatch's samples live here: this rank adds zeros to the sums
            PPO2HIP_CHECK_HIP(hipMemsetAsync(grads, 0, (size_t)P * sizeof(float), (hipStream_t)stream));
            PPO2HIP_CHECK_HIP(hipMemsetAsync(losses + 8 * j, 0, 8 * sizeof(float), (hipStream_t)stream));
        }
        if ((rc = ppo2hip_allreduce_grads(comm, grads, P, stream))) return rc;
        rc = ppo2hip_adam_step(params, grads, adam_m, adam_v, P, lr, t_first + j, beta1, beta2, epsilon, 1.0,
                               max_grad_norm, adam_scratch, stream);
        if (rc) return rc;
    }
    return PPO2HIP_OK;
}

// prx_comm / prx_allreduce_grad: the exchange step of the sharded iteration behind the C ABI.
//
// The path has ONE real exchange per step -- the SUM over ranks of dL/d(image) (786 KB at 256x256, 3 MB at 512x512) -- plus
// two scalar-sized ones for the batch-global min / max renormalisation (slip.py:21-36).  All of them are latency-bound, so
// the collective here is a ONE-SHOT DIRECT-WRITE all-reduce over IPC-mapped peer windows instead of a ring:
//
//   * every rank owns a fine-grained (uncached, system-coherent) window  [2 parities][world slots][max_bytes] + flags;
//     the windows of all peers are mapped once at connect time through hipIpc memory handles (xGMI is point-to-point: a
//     rank writes straight into each peer's HBM, 7 links in parallel on an 8-GPU node);
//   * one kernel per all-reduce: (1) each rank copies its vector into slot[rank] of EVERY window (its own included) with
//     16-byte stores, (2) a system-scope release, then the rank's sequence number goes into flag[rank] of every window,
//     (3) each rank waits until its own window shows the current sequence number in all `world` flags, (4) and sums the
//     slots IN RANK ORDER into the caller's buffer: every rank adds the same numbers in the same order, so the result is
//     bit-identical on all ranks (the decoder backward that follows is chaotic in bf16/fp16 -- ranks must not diverge)
//     and reproducible run to run;
//   * two parities of slots: a rank can be at most one call ahead of a peer (it needs that peer's flag of call k+1,
//     which the peer raises only after it finished reading call k), so call k+2 may overwrite call k's slots.
//
// Cost model on an 8-GPU MI355X node, 786 KB: 7 x 786 KB leave each GPU over 7 links (~50 GB/s effective each, in
// parallel) = ~16 us + one flag round trip, against RCCL's generic ring / tree for the same message.  Only one GPU is
// reachable from the build container, so what is TESTED is the protocol: two processes sharing one device
// (tests/test_comm_gpu.py; RCCL itself refuses two ranks on one device) and the world = 1 degenerate case; the multi-GPU
// numbers come from the driver's scaling run (bench.py prints collectives_ms_per_step).
//
// No torch types, no RCCL dependency: handles travel as opaque 64-byte blobs that the host side exchanges however it
// likes (pixray_amd/comm.py uses torch.distributed.all_gather_object once at start-up).
#include "common.h"
#include "../../include/prx.h"
#include <vector>
#include <string.h>

namespace {

constexpr int COMM_MAX_WORLD = 16;
constexpr int COMM_BLOCKS = 64;          // co-resident by construction (one block per CU at most)
constexpr int COMM_THREADS = 256;

struct CommWindowHdr {                   // lives at the start of every window
    unsigned long long flag[2][COMM_MAX_WORLD];     // [parity][writer rank] = sequence number of the data in that slot
    unsigned int arrive[2];                          // local block-arrival counters (only the owner touches them)
    unsigned int pad0[2];
    unsigned long long seq;                          // sequence number of the owner's LAST call (only the owner touches it): it lives on the
                                                     // device, not in a kernel argument, so that a call captured in a hipGraph advances on replay
    unsigned int pad[10];
};
static_assert(sizeof(CommWindowHdr) % 16 == 0, "slots must stay 16-byte aligned");

struct CommPeers { char* win[COMM_MAX_WORLD]; };

__device__ __forceinline__ void st_flag_sys(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_flag_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// slot address inside a window
__device__ __forceinline__ char* slot_of(char* win, int parity, int rank, int world, size_t max_bytes) {
    return win + sizeof(CommWindowHdr) + ((size_t)parity * world + rank) * max_bytes;
}

// OP: PRX_COMM_SUM_F32 (the image gradient), PRX_COMM_MAX_F32 (the {-min, max} pair of the batch-global renormalisation,
// slip.py:21-36) or PRX_COMM_SUM_F64 (its four backward sums); `n` counts 4-byte words, padded to a multiple of 4 by the host.
//
// CO-RESIDENCY: a block waits (step 3) for flags that its own rank raises only after ALL of this launch's blocks have arrived
// (step 2), so every block of the launch must be resident at once.  The launch is <= COMM_BLOCKS = 64 blocks of 256 threads
// with no LDS to speak of and a few dozen registers -- a quarter of the CUs at one block each, and a CU takes eight such
// blocks -- so it is resident as a whole whenever the device is not fully occupied by OTHER streams' long-running kernels;
// the Session issues it on the iteration's own stream, where nothing else runs beside it.  A launch that could not become
// resident ends in the bounded wait below, i.e. in a reported error, not in a hang.
template <int OP>
__global__ __launch_bounds__(COMM_THREADS) void oneshot_allreduce_kernel(CommPeers peers, float* __restrict__ data, size_t n, int rank, int world,
                                                                        size_t max_bytes, int* __restrict__ err) {
    // this call's sequence number: one more than the last call's, read by every block BEFORE it counts itself in below -- the
    // counter is bumped by the last block to arrive, i.e. after all of them have read it
    const unsigned long long seq = reinterpret_cast<const CommWindowHdr*>(peers.win[rank])->seq + 1;
    const int parity = (int)(seq & 1);
    const size_t n4 = n >> 2;                                  // 16-byte chunks
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    // (1) push: my vector into slot[rank] of every window
    const float4* src = reinterpret_cast<const float4*>(data);
    for (int p = 0; p < world; ++p) {
        float4* dst = reinterpret_cast<float4*>(slot_of(peers.win[(rank + p) % world], parity, rank, world, max_bytes));
        for (size_t i = tid; i < n4; i += nthr) dst[i] = src[i];
    }
    // (2) all of this block's stores are out before it counts itself in; the last block to arrive raises the flags
    __threadfence_system();
    __syncthreads();
    CommWindowHdr* mine = reinterpret_cast<CommWindowHdr*>(peers.win[rank]);
    __shared__ int last;
    __shared__ int timed_out;
    if (threadIdx.x == 0) {
        timed_out = 0;
        const unsigned prev = __hip_atomic_fetch_add(&mine->arrive[parity], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == gridDim.x - 1;
        if (last) {
            __hip_atomic_store(&mine->arrive[parity], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for call seq + 2
            mine->seq = seq;
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                st_flag_sys(&reinterpret_cast<CommWindowHdr*>(peers.win[p])->flag[parity][rank], seq);
        }
    }
    // (3) wait for every rank's data in MY window (bounded spin: a lost peer must not hang the box)
    if (threadIdx.x == 0) {
        for (int r = 0; r < world; ++r) {
            long long spins = 0;
            while (ld_flag_sys(&mine->flag[parity][r]) != seq) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1ll << 26)) { if (err) atomicExch(err, 1 + r); timed_out = 1; break; }
            }
            if (timed_out) break;
        }
        __threadfence_system();
    }
    __syncthreads();
    // (4) combine the slots in rank order (identical on every rank).  After a timed-out wait the slots are not all this call's
    // data: the result is poisoned with NaN instead of a plausible partial sum, and prx_comm_status reports the rank
    float4* out = reinterpret_cast<float4*>(data);
    if (timed_out) {
        const float q = __builtin_nanf("");
        for (size_t i = tid; i < n4; i += nthr) out[i] = make_float4(q, q, q, q);
        return;
    }
    for (size_t i = tid; i < n4; i += nthr) {
        float4 acc = reinterpret_cast<const float4*>(slot_of(peers.win[rank], parity, 0, world, max_bytes))[i];
        for (int r = 1; r < world; ++r) {
            const float4 v = reinterpret_cast<const float4*>(slot_of(peers.win[rank], parity, r, world, max_bytes))[i];
            if (OP == PRX_COMM_SUM_F32) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            else if (OP == PRX_COMM_MAX_F32) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
            else {
                double2 a = __builtin_bit_cast(double2, acc);
                const double2 b = __builtin_bit_cast(double2, v);
                a.x += b.x; a.y += b.y;
                acc = __builtin_bit_cast(float4, a);
            }
        }
        out[i] = acc;
    }
}

}  // namespace

struct prx_comm {
    int rank, world;
    size_t max_bytes, win_bytes;
    char* window;                 // this rank's window (fine-grained device memory)
    char* peer[COMM_MAX_WORLD];   // every rank's window as mapped here (peer[rank] == window)
    bool connected;
    unsigned long long seq;
    int* err;                     // device flag raised by a timed-out wait
    float* tail;                  // 4-float staging buffer for vectors whose length is not a multiple of 4 (unused when aligned)
};

extern "C" {

int prx_comm_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int prx_comm_create(prx_comm** out, int rank, int world, size_t max_bytes) {
    PRX_REQUIRE(out != nullptr, "comm_create: null out");
    PRX_REQUIRE(world >= 1 && world <= COMM_MAX_WORLD && rank >= 0 && rank < world, "comm_create: bad rank %d / world %d (max %d)", rank, world,
                COMM_MAX_WORLD);
    PRX_REQUIRE(max_bytes >= 16, "comm_create: max_bytes too small");
    prx_comm* c = new prx_comm();
    c->rank = rank; c->world = world; c->connected = false; c->seq = 0; c->err = nullptr; c->tail = nullptr;
    c->max_bytes = (max_bytes + 255) & ~(size_t)255;
    c->win_bytes = sizeof(CommWindowHdr) + 2 * (size_t)world * c->max_bytes;
    for (int i = 0; i < COMM_MAX_WORLD; ++i) c->peer[i] = nullptr;
    void* w = nullptr;
    // fine-grained: peers' stores and this rank's polls bypass the non-coherent caches (what RCCL's own low-latency
    // protocols allocate their exchange buffers as)
    hipError_t e = hipExtMallocWithFlags(&w, c->win_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { delete c; prx_set_error("comm_create: hipExtMallocWithFlags(%zu) -> %s", c->win_bytes, hipGetErrorString(e)); return -1; }
    c->window = (char*)w;
    if (hipMemset(w, 0, sizeof(CommWindowHdr)) != hipSuccess || hipMalloc((void**)&c->err, sizeof(int)) != hipSuccess ||
        hipMemset(c->err, 0, sizeof(int)) != hipSuccess) {
        (void)hipFree(w); delete c; prx_set_error("comm_create: window initialisation failed"); return -1;
    }
    c->peer[rank] = c->window;
    if (world == 1) c->connected = true;
    *out = c;
    return 0;
}

int prx_comm_export(prx_comm* c, void* handle_out) {
    PRX_REQUIRE(c && handle_out, "comm_export: null argument");
    hipIpcMemHandle_t h;
    PRX_CHECK_HIP(hipIpcGetMemHandle(&h, c->window));
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

// handles: world blobs of prx_comm_handle_bytes() bytes each, in rank order (this rank's own entry is ignored)
int prx_comm_connect(prx_comm* c, const void* handles) {
    PRX_REQUIRE(c && handles, "comm_connect: null argument");
    PRX_REQUIRE(!c->connected || c->world == 1, "comm_connect: already connected");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
        void* p = nullptr;
        PRX_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = (char*)p;
    }
    c->connected = true;
    return 0;
}

// In-place reduction over ranks on `stream`; every rank must call it with the same op and length, in the same order.
// n_words: the vector's length in 4-byte words (a double counts 2), a multiple of 4; 16-byte aligned.
int prx_allreduce(prx_comm* c, void* data, size_t n_words, int op, prx_stream_t stream) {
    PRX_REQUIRE(c && data, "allreduce: null argument");
    PRX_REQUIRE(c->connected, "allreduce: prx_comm_connect has not been called");
    PRX_REQUIRE(op == PRX_COMM_SUM_F32 || op == PRX_COMM_MAX_F32 || op == PRX_COMM_SUM_F64, "allreduce: unknown op %d", op);
    PRX_REQUIRE(n_words % 4 == 0 && ((uintptr_t)data & 15) == 0, "allreduce: the vector must be 16-byte aligned with a length multiple of 16 bytes (n_words=%zu)", n_words);
    PRX_REQUIRE(n_words * sizeof(float) <= c->max_bytes, "allreduce: %zu bytes exceed the window slot (%zu)", n_words * sizeof(float), c->max_bytes);
    if (c->world == 1) return 0;
    CommPeers peers;
    for (int i = 0; i < COMM_MAX_WORLD; ++i) peers.win[i] = c->peer[i];
    const int blocks = (int)std::min<size_t>(COMM_BLOCKS, std::max<size_t>(1, (n_words / 4 + COMM_THREADS - 1) / COMM_THREADS));
    hipStream_t st = (hipStream_t)stream;
    float* f = (float*)data;
    if (op == PRX_COMM_SUM_F32)
        hipLaunchKernelGGL(oneshot_allreduce_kernel<PRX_COMM_SUM_F32>, dim3(blocks), dim3(COMM_THREADS), 0, st, peers, f, n_words, c->rank, c->world, c->max_bytes, c->err);
    else if (op == PRX_COMM_MAX_F32)
        hipLaunchKernelGGL(oneshot_allreduce_kernel<PRX_COMM_MAX_F32>, dim3(blocks), dim3(COMM_THREADS), 0, st, peers, f, n_words, c->rank, c->world, c->max_bytes, c->err);
    else
        hipLaunchKernelGGL(oneshot_allreduce_kernel<PRX_COMM_SUM_F64>, dim3(blocks), dim3(COMM_THREADS), 0, st, peers, f, n_words, c->rank, c->world, c->max_bytes, c->err);
    PRX_LAUNCH_CHECK();
    return 0;
}
// the image-gradient all-reduce (SUM, fp32): the exchange step the north star names
int prx_allreduce_grad(prx_comm* c, float* grad, size_t n, prx_stream_t stream) { return prx_allreduce(c, grad, n, PRX_COMM_SUM_F32, stream); }

// 0 = no wait has timed out so far; r + 1 = a wait for rank r's data gave up (synchronises the device)
int prx_comm_status(prx_comm* c) {
    if (!c) return -1;
    int v = 0;
    if (hipMemcpy(&v, c->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

void prx_comm_destroy(prx_comm* c) {
    if (!c) return;
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->window) (void)hipFree(c->window);
    if (c->err) (void)hipFree(c->err);
    delete c;
}

}  // extern "C"

// tp_comm.hip — the tensor-parallel exchange step of the denoiser forward, inside the library (SURVEY.md §8b
// mmada_allreduce_init, §8e).  The reference has no tensor parallelism (inference.py:83-85 loads one replica); this is
// the one place the 8B block needs a collective: after each of the two row-parallel GEMMs (attn_out, ff_out;
// model/modeling_llada.py:741-744, 968-970) every rank holds a partial [M, d] sum.
//
// Instead of an all-reduce of the replicated residual stream followed by a replicated RMSNorm, one exchange is
//     reduce-scatter  ->  residual add + RMSNorm on the 1/tp rows this rank OWNS  ->  all-gather of the normalised rows
// so the residual stream itself is never communicated (each rank keeps only its own rows of it), the RMSNorm work is
// divided by tp, and the all-gathered tensor is exactly the A operand of the next column-parallel GEMM (or, after the
// last block, ln_f(x) for the LM head).  Bytes on the fabric are those of one all-reduce.
//
// Two transports behind the same step:
//   pull (mode 1)  every rank maps its peers' published buffers (hipIpc handles between processes, plain pointers for
//                  peers in one process) and only ever READS remote memory: the reduce kernel pulls the tp partial
//                  slices of its own rows with system-scope loads and sums them in rank order (deterministic); the
//                  gather kernel pulls the other owners' normalised rows.  Everything a GEMM reads was therefore written
//                  by a kernel of its own device.  Hand-offs are monotonic sequence counters in fine-grained memory,
//                  published by a 1-thread kernel after the producing kernel has retired and awaited by a 1-wave kernel
//                  (bounded spin: a lost peer raises an error flag, it never hangs the device).  The counters live in
//                  device memory, so the whole exchange is hipGraph-replayable.  xGMI is a full mesh of point-to-point
//                  links: the tp-1 pulls of a rank run over tp-1 different links at once (SURVEY.md §5.8 two-shot).
//   RCCL (mode 2)  ncclReduceScatter / ncclAllGather issued from here (librccl is dlopen'ed: no link-time dependency, the
//                  host may hand in the library torch already loaded), same owner-side kernel in between.
//   copy (mode 4)  round 5: the same mapped peer buffers and hand-off counters as pull, but the bytes are moved by the COPY
//                  ENGINES (hipMemcpyAsync on the mapped peer pointers: SDMA between devices): the tp-1 peer slices of this
//                  rank's rows into local staging, then the owner kernel reads LOCAL memory only; the all-gather half is
//                  tp-1 copies and no kernel.  No compute unit waits on a remote load, so nothing of the exchange competes
//                  with the GEMMs for CUs except the owner kernel (1/tp of the rows, local operands).
// CU partition (round 5, mmada_comm_set_partition): the 8-phase GEMM's 320x256 tile holds 2 waves x 256 VGPRs per SIMD and
// 144+ KiB of LDS — while one runs on a CU no exchange wave can be resident there, so "the exchange runs under the next GEMM" was
// time-slicing at workgroup granularity driven by stream priority.  With a partition the exchange stream is created with a CU
// mask of `n` CUs (bit i of the mask is CU i / 8 of XCD i % 8: the low n bits spread over all eight XCDs) and the forward's
// compute kernels run on a library stream masked to the OTHER CUs, joined to the caller's stream by events at both ends:
// the exchange kernels own their CUs by construction, the GEMMs lose n / 256 of the chip, deterministically.
// Overlap: the M rows are cut into two chunks; the exchange of chunk i runs on a second (high-priority) stream under the
// row-parallel GEMM of chunk i+1 / the next column-parallel GEMM of chunk i-1 — also at batch 1, the only join point is
// the attention (needs every key).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "handle.h"

std::atomic<int> g_allow_single_rank{0};   // mmada_set_option("tp_allow_single_rank", 1): test switch, see mmada_comm_create
void tp_allow_single_rank(int on) { g_allow_single_rank = on != 0; }

namespace {

constexpr int TP_MAX = 8;
constexpr int MAXCH = 8;  // 16-B chunks per lane: d <= 4096

constexpr int STAT_ROWS = 16384;  // text rows (B*T) a vocabulary-parallel select can take

struct TpPeers {
    const bf16_t* part[TP_MAX];
    const bf16_t* hn[TP_MAX];
    const uint32_t* ctr[TP_MAX];
    const TextStat* stats[TP_MAX];
};

struct RcclApi {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // optional (reporting only)
};

}  // namespace

struct TpComm {
    int mode = 0;  // 0: buffers allocated, not connected; 1: pull over mapped peer buffers; 2: RCCL;
                   // 3: DIAGNOSTIC "no exchange" (owner-side kernel on this rank's own partial only: wrong values, timing only)
                   // 4: copy engines over the mapped peer buffers (connected like 1)
    int rank = 0, size = 1, max_rows = 0, d = 0;
    bf16_t* part = nullptr;    // [max_rows + 8*size, d]  published: this rank's partial of the row-parallel GEMM
    bf16_t* hn_pub = nullptr;  // [max_rows + 8*size, d]  published: normalised rows this rank owns (at their global row)
    uint32_t* ctr = nullptr;   // [16] published sequence counter (fine-grained memory when the runtime grants it)
    TextStat* stats_pub = nullptr;  // [STAT_ROWS] published: this rank's per-row record of the vocabulary-parallel text head
    TextStat* stats_all = nullptr;  // [size][STAT_ROWS] RCCL transport: all-gathered records
    bf16_t* head_buf = nullptr;     // [head_rows, ceil(V/size)] this rank's logit slice (allocated at first use)
    size_t head_bytes = 0;
    uint32_t* seq = nullptr;   // [1]  private: number of hand-offs this rank has published
    int* err = nullptr;        // [1]  private: != 0 after a wait timed out (1 + the peer that never arrived)
    bool ctr_fine = false, data_fine = false;
    TpPeers peers{};
    void* opened[4][TP_MAX] = {};
    hipStream_t sc = nullptr;  // exchange stream
    hipStream_t s_cmp = nullptr;  // CU partition: compute stream masked to the CUs the exchange stream does not own (else null)
    int part_cus = 0;             // CUs of the exchange stream's mask (0: no partition)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bf16_t* stage = nullptr;   // copy transport: [size][ceil(max_rows / size) + 16, d] peer slices of this rank's rows
    size_t stage_stride = 0;   // elements per peer
    hipEvent_t ev_g[2] = {}, ev_c[2] = {};
    int chunks = 2;
    long long timeout = 0;     // hand-off timeout in wall_clock64 ticks (100 MHz)
    // RCCL
    RcclApi nccl;
    ncclComm_t comm = nullptr;
    bf16_t* rs_tmp = nullptr;  // [ceil(max_rows/size)+8, d]
};

namespace {

// ---- hand-off ------------------------------------------------------------------------------------------------------
// `seq` (private) and `ctr` (published) live in one fine-grained block and are only ever touched with system-scope atomics:
// the 1-thread kernels below land on a different XCD every time, and a plain load could be served from that XCD's L2
// (a line left there by an earlier owner of the address) instead of memory.
__global__ void tp_signal_kernel(uint32_t* seq, uint32_t* ctr) {
    const uint32_t v = __hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    __hip_atomic_store(seq, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the producing kernel retired before this one started (stream order) and each of its workgroups ended with a
    // system-scope release; this fence + drain orders the counter behind everything this wave has seen
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(ctr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void tp_wait_kernel(const uint32_t* seq, TpPeers p, int size, int rank, int* err, long long timeout_ticks) {
    const int j = threadIdx.x;
    if (j < size && j != rank && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        const uint32_t v = __hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(p.ctr[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - v) < 0) {
            __builtin_amdgcn_s_sleep(20);
            if (wall_clock64() - t0 > timeout_ticks) {  // never hang the device: flag it, results are void
                __hip_atomic_store(err, 1 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// 16 bytes of a peer's buffer at system scope (sc0 sc1: never served from this agent's caches) as ONE 16-byte request.
// A relaxed system-scope __hip_atomic_load lowers to an sc0 sc1 load only up to 8 bytes, and two of those per 16 bytes use
// half of every 64-byte fabric request each and ask for every line twice (round-2 review: 2x read amplification on xGMI).
// The buffer form carries the cache-policy bits in its aux operand (1 = sc0, 16 = sc1) and is counted by hipcc's own
// s_waitcnt bookkeeping, unlike an inline-asm load.  `base` must be wave-uniform (a kernel argument): the descriptor is
// built in SGPRs; the per-lane part is a 32-bit byte offset (mmada_comm_create refuses buffers of 4 GiB or more).
MM_DEVICE u32x4 load_sys16(const void* base, uint32_t byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffffu, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 17);
}

struct ReduceArgs {
    TpPeers p;
    int size, rank;
    int nsrc;              // pull: = size (source j = rank j's partial, the own one read locally); RCCL: 1 (pre-summed rows)
    const bf16_t* presum;  // RCCL: [r1 - r0, d] rows already summed over ranks (row 0 = global row r0)
    int src_row0[TP_MAX];  // row of source j's buffer that holds stream row 0 of this launch: 0 for a peer's whole partial buffer
                           // (pull), r0 for a staged slice whose first row is this rank's first owned row (copy transport)
    const bf16_t* part;    // this rank's own partial [*, d]
    bf16_t* x;             // residual stream [*, d]: rows [r0, r1) are this rank's, updated in place
    const bf16_t* w;       // RMSNorm weight [d]
    bf16_t* hn_pub;        // published normalised rows (global row index)
    bf16_t* xn;            // this rank's full normalised buffer: own rows are written here directly
    int r0, r1, d;
    float eps;
};

// One wave per owned row:  s = sum_j partial_j[m]  (fp32, rank order) ; out = bf16(s) ; x[m] = bf16(x[m] + out)
// (the reference: x + attn_out(att) / x + ff_out(h), model/modeling_llada.py:953, 968-970 — every nn.Linear output is
// rounded to bf16 before the residual add) ; hn = RMSLayerNorm(x[m]) (:315-329, cast-then-scale, same summation order as
// rmsnorm_row in elementwise.hip, so a row normalises to the same bits on any rank count).
// TP = compile-time rank count (0: run-time a.size): with it the peer loop unrolls and the tp-1 remote loads of a chunk are
// all in flight together — one fabric round trip per 16-byte chunk instead of one per (chunk, peer).
template <int TP>
__global__ __launch_bounds__(256) void tp_reduce_norm_kernel(ReduceArgs a) {
    const int m = a.r0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.r1) return;
    // consumer side of the hand-off (guide G16): the wait kernel saw every peer's counter, but THIS wave's caches may still
    // hold lines of the peers' buffers from the previous exchange
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const int lane = threadIdx.x & 63;
    const int nchunk = a.d >> 3;
    u32x4 xv[MAXCH];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c >= nchunk) break;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (a.presum) {
            const u32x4 v = ((const u32x4*)(a.presum + (size_t)(m - a.r0) * a.d))[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] = __uint_as_float(v[e] << 16);
                acc[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
            }
        } else {
            if constexpr (TP > 0) {
                u32x4 pv[TP];
#pragma unroll
                for (int j = 0; j < TP; ++j)  // p.part[rank] is this rank's own buffer: one uniform load form, no branch
                    pv[j] = load_sys16(a.p.part[j], ((uint32_t)(m - a.src_row0[j]) * (uint32_t)a.d + c * 8) * 2u);
#pragma unroll
                for (int j = 0; j < TP; ++j)  // rank order: the sum is the same on whichever rank owns the row
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] += __uint_as_float(pv[j][e] << 16);
                        acc[2 * e + 1] += __uint_as_float(pv[j][e] & 0xffff0000u);
                    }
            } else {
                for (int j = 0; j < a.size; ++j) {
                    const u32x4 v = (j == a.rank) ? ((const u32x4*)(a.part + (size_t)m * a.d))[c]
                                                  : load_sys16(a.p.part[j], ((uint32_t)(m - a.src_row0[j]) * (uint32_t)a.d + c * 8) * 2u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] += __uint_as_float(v[e] << 16);
                        acc[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u);
                    }
                }
            }
        }
        const u32x4 r = ((const u32x4*)(a.x + (size_t)m * a.d))[c];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = bfround(__uint_as_float(r[e] << 16) + bfround(acc[2 * e]));
            const float hi = bfround(__uint_as_float(r[e] & 0xffff0000u) + bfround(acc[2 * e + 1]));
            ss += lo * lo;
            ss += hi * hi;
            o[e] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
        }
        xv[i] = o;
        ((u32x4*)(a.x + (size_t)m * a.d))[c] = o;
    }
    ss = wave_sum(ss);
    const float var = ss / (float)a.d;
    const float rs = 1.0f / sqrtf(var + a.eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c >= nchunk) break;
        const u32x4 wv = ((const u32x4*)a.w)[c];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(xv[i][e] << 16), hi = __uint_as_float(xv[i][e] & 0xffff0000u);
            const float wl = __uint_as_float(wv[e] << 16), wh = __uint_as_float(wv[e] & 0xffff0000u);
            o[e] = pack_bf2(wl * bfround(lo * rs), wh * bfround(hi * rs));
        }
        ((u32x4*)(a.hn_pub + (size_t)m * a.d))[c] = o;
        ((u32x4*)(a.xn + (size_t)m * a.d))[c] = o;
    }
    // producer side: the published row must have left this XCD's L2 before the counter that follows this kernel moves
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// rows [m0, m1) owned by other ranks: pull them from their owner's published buffer into dst (one wave per row)
__global__ __launch_bounds__(256) void tp_gather_kernel(TpPeers p, int rank, int m0, int m1, int slice, int d,
                                                        bf16_t* dst, int use_part) {
    const int m = m0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= m1) return;
    const int owner = (m - m0) / slice;
    if (owner == rank) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // see tp_reduce_norm_kernel
    // one row per wave: the owner IS wave-uniform, but it derives from threadIdx and the compiler cannot prove it — say so,
    // so that the buffer descriptor is built once in SGPRs (no per-lane waterfall loop around every load, guide T20)
    const int owner_u = __builtin_amdgcn_readfirstlane(owner);
    const bf16_t* src = use_part ? p.part[owner_u] : p.hn[owner_u];
    const int nchunk = d >> 3;
#pragma unroll 8
    for (int c = threadIdx.x & 63; c < nchunk; c += 64)
        ((u32x4*)(dst + (size_t)m * d))[c] = load_sys16(src, ((uint32_t)m * (uint32_t)d + c * 8) * 2u);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* src, bf16_t* dst, int r0, int r1, int d) {
    const int m = r0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= r1) return;
    for (int c = threadIdx.x & 63; c < (d >> 3); c += 64) ((u32x4*)(dst + (size_t)m * d))[c] = ((const u32x4*)(src + (size_t)m * d))[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // dst may be a published buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// out[r] = src[b*Lp + l] for rows[r] = b*L + l (LM-head rows of an already normalised stream)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* src, const int32_t* rows, int R, int L, int Lp, int d,
                                                          int nflat, bf16_t* out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int flat = min(max(rows[r], 0), nflat - 1);
    const int b = flat / L, l = flat - b * L;
    const u32x4* s = (const u32x4*)(src + ((size_t)b * Lp + l) * d);
    for (int c = threadIdx.x & 63; c < (d >> 3); c += 64) ((u32x4*)(out + (size_t)r * d))[c] = s[c];
}

// Vocabulary-parallel text head: combine the tp per-rank records of every row into the conf (fp64 soft-max probability of
// the arg-max, generators/parallel_generator.py:185-205) and x0 the one-rank kernel writes.  One thread per row.
__global__ void tp_text_combine_kernel(TpPeers p, int size, int rank, const TextStat* own, const TextStat* gathered,
                                       int stat_stride, int R, double* conf_out, int32_t* x0_out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    if (!gathered) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    TextStat st[TP_MAX];
    for (int j = 0; j < size; ++j) {
        if (gathered) st[j] = gathered[(size_t)j * stat_stride + row];
        else if (j == rank) st[j] = own[row];
        else {
            const u32x4 v = load_sys16(p.stats[j], (uint32_t)row * 16u);  // one 16-byte record
            st[j].lmax = __uint_as_float(v[0]);
            st[j].arg = (int32_t)v[1];
            st[j].sum = __longlong_as_double((long long)(((uint64_t)v[3] << 32) | v[2]));
        }
    }
    float mx = -INFINITY;
    int arg = 0;
    for (int j = 0; j < size; ++j)
        if (st[j].lmax > mx) { mx = st[j].lmax; arg = st[j].arg; }  // strict >: the lowest rank (lowest column) wins a tie
    if (!(mx > -INFINITY)) {  // not a masked position (or an all -inf row)
        conf_out[row] = -INFINITY;
        x0_out[row] = 0;
        return;
    }
    double tot = 0.0;
    for (int j = 0; j < size; ++j)
        if (st[j].lmax > -INFINITY) tot += st[j].sum * exp((double)st[j].lmax - (double)mx);
    conf_out[row] = 1.0 / tot;  // exp(l[x0] - max) / sum with x0 the arg-max
    x0_out[row] = arg;
}

// One wave on (at least) every XCD writes that XCD's L2 back: for partials that were NOT produced by a publishing kernel
// of this library (mmada_comm_exchange: the caller filled the buffer with its own kernels).  Workgroup b runs on XCD b % 8.
__global__ void tp_flush_kernel() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

long long default_timeout_ticks() {
    const char* e = getenv("MMADA_TP_TIMEOUT_S");
    const double sec = e ? atof(e) : 20.0;
    return (long long)(sec * 100e6);  // wall_clock64 runs at 100 MHz on gfx9
}

int signal_wait(TpComm* c, hipStream_t s) {
    hipLaunchKernelGGL(tp_signal_kernel, dim3(1), dim3(1), 0, s, c->seq, c->ctr);
    hipLaunchKernelGGL(tp_wait_kernel, dim3(1), dim3(64), 0, s, c->seq, c->peers, c->size, c->rank, c->err, c->timeout);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

struct Slice { int m0, m1, slice, r0, r1; };

// chunk k of `nch` over M rows; every chunk but the last is a multiple of 8*tp rows, so only the last one is padded
Slice chunk_slice(int M, int tp, int rank, int nch, int k) {
    Slice s;
    const int unit = 8 * tp;
    const int first = nch == 2 ? (M / 2 + unit - 1) / unit * unit : M;
    s.m0 = k == 0 ? 0 : min(first, M);
    s.m1 = (k == nch - 1) ? M : min(first, M);
    const int rows = s.m1 - s.m0;
    s.slice = max(8, ((rows + tp - 1) / tp + 7) / 8 * 8);
    s.r0 = min(s.m1, s.m0 + rank * s.slice);
    s.r1 = min(s.m1, s.r0 + s.slice);
    return s;
}

int nccl_fail(TpComm* c, const char* what, ncclResult_t r) {
    return mm_fail("%s: %s", what, c->nccl.GetErrorString ? c->nccl.GetErrorString(r) : "RCCL error");
}

// One exchange over rows [m0, m1): partial sums in c->part  ->  x (own rows), xn (all rows) ; on stream s
int exchange(mmada_handle* h, const Slice& sl, const bf16_t* norm_w, hipStream_t s) {
    TpComm* c = h->tp;
    const int d = h->cfg.d_model;
    ReduceArgs a{};
    a.p = c->peers; a.size = c->size; a.rank = c->rank;
    a.part = c->part; a.x = h->x; a.w = norm_w; a.hn_pub = c->hn_pub; a.xn = h->xn;
    a.r0 = sl.r0; a.r1 = sl.r1; a.d = d; a.eps = h->cfg.rms_eps;
    const int own = sl.r1 - sl.r0;
    if (c->mode == 1) {
        if (signal_wait(c, s)) return 1;  // every rank's partial of this chunk is complete
        a.nsrc = c->size;
        if (own > 0) {
            const dim3 grid((own + 3) / 4), blk(256);
            switch (c->size) {
                case 2: hipLaunchKernelGGL(tp_reduce_norm_kernel<2>, grid, blk, 0, s, a); break;
                case 4: hipLaunchKernelGGL(tp_reduce_norm_kernel<4>, grid, blk, 0, s, a); break;
                case 8: hipLaunchKernelGGL(tp_reduce_norm_kernel<8>, grid, blk, 0, s, a); break;
                default: hipLaunchKernelGGL(tp_reduce_norm_kernel<0>, grid, blk, 0, s, a);
            }
        }
        if (signal_wait(c, s)) return 1;  // every owner's normalised rows are published
        hipLaunchKernelGGL(tp_gather_kernel, dim3((sl.m1 - sl.m0 + 3) / 4), dim3(256), 0, s, c->peers, c->rank, sl.m0, sl.m1,
                           sl.slice, d, h->xn, 0);
        MM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (c->mode == 4) {
        if (signal_wait(c, s)) return 1;  // every rank's partial of this chunk is complete
        // the tp - 1 peer slices of MY rows -> local staging, by the copy engines; the kernel then sums local operands
        // (src_row0[j] = r0: staging row 0 of peer j is stream row r0)
        ReduceArgs b = a;
        b.nsrc = c->size;
        if (own > 0) {
            for (int j = 0; j < c->size; ++j) {
                if (j == c->rank) { b.p.part[j] = c->part; continue; }
                bf16_t* dst = c->stage + (size_t)j * c->stage_stride;
                MM_CHECK_HIP(hipMemcpyAsync(dst, c->peers.part[j] + (size_t)sl.r0 * d, (size_t)own * d * 2, hipMemcpyDefault, s));
                b.p.part[j] = dst;
                b.src_row0[j] = sl.r0;
            }
            const dim3 grid((own + 3) / 4), blk(256);
            switch (c->size) {
                case 2: hipLaunchKernelGGL(tp_reduce_norm_kernel<2>, grid, blk, 0, s, b); break;
                case 4: hipLaunchKernelGGL(tp_reduce_norm_kernel<4>, grid, blk, 0, s, b); break;
                case 8: hipLaunchKernelGGL(tp_reduce_norm_kernel<8>, grid, blk, 0, s, b); break;
                default: hipLaunchKernelGGL(tp_reduce_norm_kernel<0>, grid, blk, 0, s, b);
            }
            MM_CHECK_HIP(hipGetLastError());
        }
        if (signal_wait(c, s)) return 1;  // every owner's normalised rows are published
        for (int j = 0; j < c->size; ++j) {   // all-gather half: tp - 1 copies, no kernel
            if (j == c->rank) continue;
            const int j0 = min(sl.m1, sl.m0 + j * sl.slice), j1 = min(sl.m1, j0 + sl.slice);
            if (j1 > j0)
                MM_CHECK_HIP(hipMemcpyAsync(h->xn + (size_t)j0 * d, c->peers.hn[j] + (size_t)j0 * d, (size_t)(j1 - j0) * d * 2,
                                            hipMemcpyDefault, s));
        }
        return 0;
    }
    if (c->mode == 3) {  // diagnostic: the forward without its exchange (bench.py: exposed exchange time = real - this)
        a.nsrc = 1; a.presum = c->part + (size_t)sl.r0 * d;
        if (own > 0) hipLaunchKernelGGL(tp_reduce_norm_kernel<0>, dim3((own + 3) / 4), dim3(256), 0, s, a);
        MM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (c->mode == 2) {
        const size_t cnt = (size_t)sl.slice * d;
        ncclResult_t r = c->nccl.ReduceScatter(c->part + (size_t)sl.m0 * d, c->rs_tmp, cnt, ncclBfloat16, ncclSum, c->comm, s);
        if (r != ncclSuccess) return nccl_fail(c, "ncclReduceScatter", r);
        a.nsrc = 1; a.presum = c->rs_tmp;
        if (own > 0) hipLaunchKernelGGL(tp_reduce_norm_kernel<0>, dim3((own + 3) / 4), dim3(256), 0, s, a);
        MM_CHECK_HIP(hipGetLastError());
        // every rank contributes its `slice` rows (rows past M are pad rows of the buffers: never read by a GEMM)
        r = c->nccl.AllGather(c->hn_pub + (size_t)(sl.m0 + c->rank * sl.slice) * d, h->xn + (size_t)sl.m0 * d, cnt,
                              ncclBfloat16, c->comm, s);
        if (r != ncclSuccess) return nccl_fail(c, "ncclAllGather", r);
        return 0;
    }
    return mm_fail("tensor-parallel forward: no transport connected (mmada_comm_connect_ipc / _local / _rccl)");
}

int load_rccl(RcclApi* api, const char* path) {
    if (api->dl) return 0;
    const char* cand[3] = {path && path[0] ? path : nullptr, "librccl.so", "librccl.so.1"};
    for (const char* p : cand) {
        if (!p) continue;
        api->dl = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
        if (api->dl) break;
    }
    if (!api->dl) return mm_fail("RCCL: cannot dlopen librccl (%s)", dlerror());
#define LOAD(field, sym)                                                            \
    api->field = (decltype(api->field))dlsym(api->dl, sym);                        \
    if (!api->field) return mm_fail("RCCL: symbol %s missing", sym)
    LOAD(GetUniqueId, "ncclGetUniqueId");
    LOAD(CommInitRank, "ncclCommInitRank");
    LOAD(CommDestroy, "ncclCommDestroy");
    LOAD(ReduceScatter, "ncclReduceScatter");
    LOAD(AllGather, "ncclAllGather");
    LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
    api->CommCount = (decltype(api->CommCount))dlsym(api->dl, "ncclCommCount");
    return 0;
}

struct CommExport {  // what a rank hands its peers (mmada_comm_create -> mmada_comm_connect_ipc): 4 x 64 B
    hipIpcMemHandle_t part, hn, ctr, stats;
};

}  // namespace

// ---- forward ---------------------------------------------------------------------------------------------------------
// All blocks of a tensor-parallel forward.  h->x holds the embeddings (replicated), every rank keeps its own rows of the
// residual stream from here on.  Compute on `s`, exchanges on the library's second stream.
static int tp_forward_body_on(mmada_handle* h, hipStream_t s);

int tp_forward_body(mmada_handle* h, hipStream_t s_user) {
    TpComm* c = h->tp;
    if (!c || c->mode == 0) return mm_fail("tensor-parallel forward: no transport connected (mmada_comm_create + connect)");
    if (!c->s_cmp) return tp_forward_body_on(h, s_user);
    // CU partition: the blocks run on the library's masked compute stream, forked from and joined to the caller's stream
    MM_CHECK_HIP(hipEventRecord(c->ev_in, s_user));
    MM_CHECK_HIP(hipStreamWaitEvent(c->s_cmp, c->ev_in, 0));
    const int rc = tp_forward_body_on(h, c->s_cmp);
    MM_CHECK_HIP(hipEventRecord(c->ev_out, c->s_cmp));
    MM_CHECK_HIP(hipStreamWaitEvent(s_user, c->ev_out, 0));
    return rc;
}

static int tp_forward_body_on(mmada_handle* h, hipStream_t s) {
    TpComm* c = h->tp;
    if (h->M > c->max_rows) return mm_fail("tensor-parallel forward: %d rows exceed the comm buffers (%d)", h->M, c->max_rows);
    const int d = h->cfg.d_model, tp = c->size, M = h->M, nl = h->cfg.n_layers;
    if ((d >> 3) > 64 * MAXCH) return mm_fail("tensor-parallel forward: d_model > %d is not supported", 64 * MAXCH * 8);
    const int nch = (c->chunks >= 2 && M >= 4 * 8 * tp) ? 2 : 1;
    Slice sl[2];
    for (int k = 0; k < nch; ++k) sl[k] = chunk_slice(M, tp, c->rank, nch, k);
    const double rows_real = (double)h->B * h->L / M;  // fraction of stream rows that are not padding (FLOP accounting)
    const CacheSlot* cc = h->cc;
    // first RMSNorm of the forward: the embeddings are replicated, no exchange needed
    if (h->xn_is_layer0) h->xn_is_layer0 = false;  // fused into the embedding kernel
    else if (launch_rmsnorm(h->x, h->layers[0].attn_norm, h->xn, M, d, h->cfg.rms_eps, s)) return 1;
    bool pending[2] = {false, false};  // chunk k's xn rows are being produced on the exchange stream
    auto after_gemm_exchange = [&](int k, const bf16_t* w) -> int {
        MM_CHECK_HIP(hipEventRecord(c->ev_g[k], s));
        MM_CHECK_HIP(hipStreamWaitEvent(c->sc, c->ev_g[k], 0));
        if (exchange(h, sl[k], w, c->sc)) return 1;
        MM_CHECK_HIP(hipEventRecord(c->ev_c[k], c->sc));
        pending[k] = true;
        return 0;
    };
    auto need_xn = [&](int k) -> int {
        if (pending[k]) {
            MM_CHECK_HIP(hipStreamWaitEvent(s, c->ev_c[k], 0));
            pending[k] = false;
        }
        return 0;
    };
    for (int layer = 0; layer < nl; ++layer) {
        const LayerWeights& lw = h->layers[layer];
        // ---- q/k/v (column-parallel: this rank's heads), RoPE in the epilogue ----
        for (int k = 0; k < nch; ++k) {
            if (need_xn(k)) return 1;
            GemmArgs g{};
            g.A = h->xn + (size_t)sl[k].m0 * d; g.W = lw.wqkv; g.C = nullptr;
            g.M = sl[k].m1 - sl[k].m0; g.N = (h->hq_l + 2 * h->hkv_l) * 128; g.K = d;
            g.lda = d; g.ldw = d; g.ldc = 0; g.m_base = sl[k].m0;
            g.q = h->q; g.k = h->k; g.vT = h->vT; g.rope_cos = h->rope_cos; g.rope_sin = h->rope_sin;
            g.Lp = h->Lp; g.Lkv = h->Lkv; g.Hq = h->hq_l; g.Hkv = h->hkv_l;
            if (cc) {   // dLLM cache step (mmada_forward_cached): this rank's heads of the block's keys / values live in the slot
                g.k = cc->K(layer); g.vT = cc->vT(layer); g.Lkv = cc->Lkv;
                g.pos_map = h->cc_pos; g.Lq = h->Lkv; g.q_pos_shift = h->cc_qshift;
            }
            ProfScope p(h, layer, 0, 2.0 * g.M * rows_real * g.N * g.K, s);
            if (launch_gemm(EPI_QKV, g, s)) return 1;
        }
        {   // ---- attention over this rank's heads: the one join point (every key of a sequence) ----
            ProfScope p(h, layer, 1, 4.0 * h->hq_l * (double)h->B * h->L * (cc ? cc->L : h->L) * 128.0, s);
            if (cc) {   // compact (or all) queries of this call against the slot's keys / values of the whole sequence
                if (launch_attention(h->q, cc->K(layer), cc->vT(layer), h->att, h->B, h->hq_l, h->hkv_l, cc->L, h->Lp, cc->Lkv,
                                     h->Lp, h->hq_l * 128, s, 0, h->Lkv)) return 1;
            } else if (launch_attention(h->q, h->k, h->vT, h->att, h->B, h->hq_l, h->hkv_l, h->L, h->Lp, h->Lkv, h->Lp,
                                        h->hq_l * 128, s)) return 1;
        }
        // ---- attn_out (row-parallel) chunk by chunk; chunk k's exchange runs under chunk k+1's GEMM ----
        for (int k = 0; k < nch; ++k) {
            GemmArgs o{};
            o.A = h->att + (size_t)sl[k].m0 * h->hq_l * 128; o.W = lw.wo; o.C = c->part + (size_t)sl[k].m0 * d;
            o.M = sl[k].m1 - sl[k].m0; o.N = d; o.K = h->hq_l * 128;
            o.lda = o.K; o.ldw = o.K; o.ldc = d; o.publish = c->mode == 1 || c->mode == 4;
            {
                ProfScope p(h, layer, 2, 2.0 * o.M * rows_real * o.N * o.K, s);
                if (launch_gemm(EPI_STORE, o, s)) return 1;
            }
            if (after_gemm_exchange(k, lw.ff_norm)) return 1;
        }
        // ---- gate/up (column-parallel) + SiLU*mul, then down (row-parallel) ----
        for (int k = 0; k < nch; ++k) {
            if (need_xn(k)) return 1;
            GemmArgs g{};
            g.A = h->xn + (size_t)sl[k].m0 * d; g.W = lw.wgu; g.C = h->hbuf + (size_t)sl[k].m0 * h->f_l;
            g.M = sl[k].m1 - sl[k].m0; g.N = 2 * h->f_l; g.K = d;
            g.lda = d; g.ldw = d; g.ldc = h->f_l;
            ProfScope p(h, layer, 3, 2.0 * g.M * rows_real * g.N * g.K, s);
            if (launch_gemm(EPI_SWIGLU, g, s)) return 1;
        }
        const bf16_t* next_w = layer + 1 < nl ? h->layers[layer + 1].attn_norm : h->ln_f;
        for (int k = 0; k < nch; ++k) {
            GemmArgs o{};
            o.A = h->hbuf + (size_t)sl[k].m0 * h->f_l; o.W = lw.wdown; o.C = c->part + (size_t)sl[k].m0 * d;
            o.M = sl[k].m1 - sl[k].m0; o.N = d; o.K = h->f_l;
            o.lda = h->f_l; o.ldw = h->f_l; o.ldc = d; o.publish = c->mode == 1 || c->mode == 4;
            {
                ProfScope p(h, layer, 4, 2.0 * o.M * rows_real * o.N * o.K, s);
                if (launch_gemm(EPI_STORE, o, s)) return 1;
            }
            if (after_gemm_exchange(k, next_w)) return 1;
        }
    }
    for (int k = 0; k < nch; ++k)
        if (need_xn(k)) return 1;
    h->xn_is_final = true;  // xn = ln_f(x) on every row
    return 0;
}

// Residual stream of every owner -> full [M, d] (parity taps; the forward itself never moves the residual stream)
int tp_gather_stream(mmada_handle* h, bf16_t* full_out, hipStream_t s) {
    TpComm* c = h->tp;
    if (!c || c->mode == 0 || c->mode == 3) return mm_fail("tp_gather_stream: no transport connected (or the no-exchange diagnostic is on)");
    const int d = h->cfg.d_model, M = h->M;
    const int nch = (c->chunks >= 2 && M >= 4 * 8 * c->size) ? 2 : 1;
    for (int k = 0; k < nch; ++k) {
        const Slice sl = chunk_slice(M, c->size, c->rank, nch, k);
        const int own = sl.r1 - sl.r0;
        if (own > 0) {
            hipLaunchKernelGGL(copy_rows_kernel, dim3((own + 3) / 4), dim3(256), 0, s, h->x, c->hn_pub, sl.r0, sl.r1, d);
            hipLaunchKernelGGL(copy_rows_kernel, dim3((own + 3) / 4), dim3(256), 0, s, h->x, full_out, sl.r0, sl.r1, d);
        }
        if (c->mode == 1 || c->mode == 4) {
            if (signal_wait(c, s)) return 1;
            hipLaunchKernelGGL(tp_gather_kernel, dim3((sl.m1 - sl.m0 + 3) / 4), dim3(256), 0, s, c->peers, c->rank, sl.m0,
                               sl.m1, sl.slice, d, full_out, 0);
            if (signal_wait(c, s)) return 1;  // hn_pub may be overwritten only after every peer has pulled
        } else {
            ncclResult_t r = c->nccl.AllGather(c->hn_pub + (size_t)(sl.m0 + c->rank * sl.slice) * d,
                                               full_out + (size_t)sl.m0 * d, (size_t)sl.slice * d, ncclBfloat16, c->comm, s);
            if (r != ncclSuccess) return nccl_fail(c, "ncclAllGather", r);
        }
        MM_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

bool tp_comm_connected(const mmada_handle* h) { return h->tp && h->tp->mode != 0; }

void tp_comm_free(mmada_handle* h) {
    TpComm* c = h->tp;
    if (!c) return;
    for (int j = 0; j < TP_MAX; ++j)
        for (int b = 0; b < 4; ++b)
            if (c->opened[b][j]) (void)hipIpcCloseMemHandle(c->opened[b][j]);
    if (c->comm && c->nccl.CommDestroy) (void)c->nccl.CommDestroy(c->comm);
    for (int k = 0; k < 2; ++k) {
        if (c->ev_g[k]) (void)hipEventDestroy(c->ev_g[k]);
        if (c->ev_c[k]) (void)hipEventDestroy(c->ev_c[k]);
    }
    if (c->sc) (void)hipStreamDestroy(c->sc);
    if (c->s_cmp) (void)hipStreamDestroy(c->s_cmp);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    (void)hipFree(c->stage);
    (void)hipFree(c->part); (void)hipFree(c->hn_pub); (void)hipFree(c->ctr);
    (void)hipFree(c->rs_tmp); (void)hipFree(c->stats_pub); (void)hipFree(c->stats_all); (void)hipFree(c->head_buf);
    delete c;
    h->tp = nullptr;
}

extern "C" {

int mmada_comm_export_bytes(void) { return (int)sizeof(CommExport); }

int mmada_comm_create(mmada_handle* h, int max_rows, void* export_out) {
    if (!h || max_rows <= 0) return mm_fail("mmada_comm_create: bad argument");
    // tp_size == 1 behind mmada_set_option("tp_allow_single_rank", 1): a one-rank group runs EVERY line of the exchange (RCCL
    // reduce-scatter / all-gather of one rank are copies, the pull transport has no peer to wait for) and must reproduce the
    // plain forward bit for bit — the test that executes the RCCL transport without a second GPU (tests/test_gpu_tp.py)
    if ((h->cfg.tp_size < 2 && !(h->cfg.tp_size == 1 && g_allow_single_rank)) || h->cfg.tp_size > TP_MAX)
        return mm_fail("mmada_comm_create: tp_size must be 2..%d", TP_MAX);
    if (h->tp) tp_comm_free(h);
    TpComm* c = new TpComm();
    c->rank = h->cfg.tp_rank; c->size = h->cfg.tp_size; c->d = h->cfg.d_model;
    c->max_rows = max_rows;
    const size_t rows = (size_t)max_rows + 8 * c->size;
    if (rows * (size_t)h->cfg.d_model * 2 >= (1ull << 32))
        return mm_fail("mmada_comm_create: %zu rows x %d exceed the 4 GiB the pull kernels address with 32-bit offsets", rows, h->cfg.d_model);
    const size_t bytes = rows * c->d * 2;
    // The hand-off counters and the 16-byte text records are read by other agents while their owner keeps writing them:
    // fine-grained (coherent) device memory when the runtime grants it.  The two bandwidth-critical buffers (partials,
    // normalised rows) stay ordinary allocations — a GEMM epilogue's 2-byte stores must combine in L2 — and are handed
    // over by the release / acquire fences of their producers and consumers; MMADA_TP_FINE_DATA=1 makes them fine-grained too.
    const char* fd_env = getenv("MMADA_TP_FINE_DATA");
    const bool fine_data_wanted = fd_env && fd_env[0] == '1';
    auto alloc_pub = [&](void** p, size_t n, bool want_fine, bool* fine) -> hipError_t {
        if (want_fine && hipExtMallocWithFlags(p, n, hipDeviceMallocFinegrained) == hipSuccess) {
            if (fine) *fine = true;
            return hipSuccess;
        }
        (void)hipGetLastError();
        if (fine) *fine = false;
        return hipMalloc(p, n);
    };
    bool fine_data = false;
    MM_CHECK_HIP(alloc_pub((void**)&c->part, bytes, fine_data_wanted, &fine_data));
    MM_CHECK_HIP(alloc_pub((void**)&c->hn_pub, bytes, fine_data_wanted, nullptr));
    MM_CHECK_HIP(hipMemset(c->part, 0, bytes));
    MM_CHECK_HIP(hipMemset(c->hn_pub, 0, bytes));
    // counters: [0] published sequence number, [64] private sequence number, [128] error flag — 4 KiB of their own
    MM_CHECK_HIP(alloc_pub((void**)&c->ctr, 4096, true, &c->ctr_fine));
    MM_CHECK_HIP(hipMemset(c->ctr, 0, 4096));
    c->seq = c->ctr + 64;
    c->err = (int*)(c->ctr + 128);
    c->data_fine = fine_data;
    MM_CHECK_HIP(hipMalloc(&c->rs_tmp, ((size_t)(max_rows + c->size - 1) / c->size + 16) * c->d * 2));
    c->stage_stride = ((size_t)(max_rows + c->size - 1) / c->size + 16) * c->d;   // one owner slice of the largest chunk
    c->stage = nullptr;  // the copy transport's landing zone: allocated by the first mmada_comm_set_mode(4)
    MM_CHECK_HIP(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    MM_CHECK_HIP(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    MM_CHECK_HIP(alloc_pub((void**)&c->stats_pub, (size_t)STAT_ROWS * sizeof(TextStat), true, nullptr));
    MM_CHECK_HIP(hipMemset(c->stats_pub, 0, (size_t)STAT_ROWS * sizeof(TextStat)));
    MM_CHECK_HIP(hipMalloc(&c->stats_all, (size_t)c->size * STAT_ROWS * sizeof(TextStat)));
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // the exchange must not queue behind a whole round of GEMM workgroups
    MM_CHECK_HIP(hipStreamCreateWithPriority(&c->sc, hipStreamNonBlocking, hi));
    for (int k = 0; k < 2; ++k) {
        MM_CHECK_HIP(hipEventCreateWithFlags(&c->ev_g[k], hipEventDisableTiming));
        MM_CHECK_HIP(hipEventCreateWithFlags(&c->ev_c[k], hipEventDisableTiming));
    }
    const char* e = getenv("MMADA_TP_CHUNKS");
    c->chunks = e ? atoi(e) : 2;
    c->timeout = default_timeout_ticks();
    MM_CHECK_HIP(hipDeviceSynchronize());
    h->tp = c;
    if (export_out) {
        CommExport ex;
        memset(&ex, 0, sizeof(ex));
        hipError_t e1 = hipIpcGetMemHandle(&ex.part, c->part), e2 = hipIpcGetMemHandle(&ex.hn, c->hn_pub),
                   e3 = hipIpcGetMemHandle(&ex.ctr, c->ctr);
        if (e3 == hipSuccess) e3 = hipIpcGetMemHandle(&ex.stats, c->stats_pub);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
            (void)hipGetLastError();
            memset(export_out, 0, sizeof(ex));
            return mm_fail("mmada_comm_create: hipIpcGetMemHandle failed (%s): peers in other processes cannot map this rank; "
                           "use mmada_comm_connect_rccl", hipGetErrorString(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3)));
        }
        memcpy(export_out, &ex, sizeof(ex));
    }
    return 0;
}

int mmada_comm_connect_ipc(mmada_handle* h, const void* exports) {
    if (!h || !h->tp || !exports) return mm_fail("mmada_comm_connect_ipc: call mmada_comm_create first");
    TpComm* c = h->tp;
    const CommExport* ex = (const CommExport*)exports;
    for (int j = 0; j < c->size; ++j) {
        if (j == c->rank) {
            c->peers.part[j] = c->part; c->peers.hn[j] = c->hn_pub; c->peers.ctr[j] = c->ctr; c->peers.stats[j] = c->stats_pub;
            continue;
        }
        void* p[4] = {nullptr, nullptr, nullptr, nullptr};
        const hipIpcMemHandle_t* hd[4] = {&ex[j].part, &ex[j].hn, &ex[j].ctr, &ex[j].stats};
        for (int b = 0; b < 4; ++b) {
            hipError_t e = hipIpcOpenMemHandle(&p[b], *hd[b], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return mm_fail("mmada_comm_connect_ipc: hipIpcOpenMemHandle(rank %d, buffer %d): %s", j, b, hipGetErrorString(e));
            }
            c->opened[b][j] = p[b];
        }
        c->peers.part[j] = (const bf16_t*)p[0]; c->peers.hn[j] = (const bf16_t*)p[1]; c->peers.ctr[j] = (const uint32_t*)p[2];
        c->peers.stats[j] = (const TextStat*)p[3];
    }
    c->mode = 1;
    return 0;
}

int mmada_comm_connect_local(mmada_handle* h, mmada_handle* const* ranks) {
    if (!h || !h->tp || !ranks) return mm_fail("mmada_comm_connect_local: call mmada_comm_create first");
    TpComm* c = h->tp;
    for (int j = 0; j < c->size; ++j) {
        if (!ranks[j] || !ranks[j]->tp) return mm_fail("mmada_comm_connect_local: rank %d has no comm", j);
        if (ranks[j]->cfg.tp_rank != j) return mm_fail("mmada_comm_connect_local: handle %d is tp_rank %d", j, ranks[j]->cfg.tp_rank);
        c->peers.part[j] = ranks[j]->tp->part; c->peers.hn[j] = ranks[j]->tp->hn_pub; c->peers.ctr[j] = ranks[j]->tp->ctr;
        c->peers.stats[j] = ranks[j]->tp->stats_pub;
    }
    c->mode = 1;
    return 0;
}

int mmada_comm_unique_id(void* out128, const char* librccl_path) {
    if (!out128) return mm_fail("mmada_comm_unique_id: null argument");
    static RcclApi api;
    if (load_rccl(&api, librccl_path)) return 1;
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return mm_fail("ncclGetUniqueId: %s", api.GetErrorString(r));
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int mmada_comm_connect_rccl(mmada_handle* h, const void* unique_id128, const char* librccl_path) {
    if (!h || !h->tp || !unique_id128) return mm_fail("mmada_comm_connect_rccl: call mmada_comm_create first");
    TpComm* c = h->tp;
    if (load_rccl(&c->nccl, librccl_path)) return 1;
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    ncclResult_t r = c->nccl.CommInitRank(&c->comm, c->size, id, c->rank);
    if (r != ncclSuccess) return nccl_fail(c, "ncclCommInitRank", r);
    c->mode = 2;
    MM_CHECK_HIP(hipMemset(c->err, 0, sizeof(int)));  // a failed attempt with the pull transport must not stick to this one
    return 0;
}

/* Hand-off timeout of the pull transport in seconds (<= 0: back to MMADA_TP_TIMEOUT_S / 20 s); also clears a sticky error.
 * A start-up self-test uses a short one so that a transport that cannot work is abandoned quickly. */
int mmada_comm_set_timeout(mmada_handle* h, double seconds) {
    if (!h || !h->tp) return mm_fail("mmada_comm_set_timeout: no comm");
    h->tp->timeout = seconds > 0 ? (long long)(seconds * 100e6) : default_timeout_ticks();
    MM_CHECK_HIP(hipDeviceSynchronize());
    MM_CHECK_HIP(hipMemset(h->tp->err, 0, sizeof(int)));
    return 0;
}

/* mode: 0 none, 1 pull (IPC / same-process peers), 2 RCCL, 3 the "no exchange" DIAGNOSTIC (the forward's values are void: callers
 * that report results must treat 3 as an error — check_tp_exchange does).  err: != 0 after a hand-off timed out (synchronises
 * `stream`). */
int mmada_comm_status(mmada_handle* h, int* mode_out, int* err_out, int* finegrained_out, void* stream) {
    if (!h) return mm_fail("mmada_comm_status: null handle");
    if (mode_out) *mode_out = h->tp ? h->tp->mode : 0;
    if (finegrained_out) *finegrained_out = h->tp ? ((int)h->tp->ctr_fine | ((int)h->tp->data_fine << 1)) : 0;
    if (err_out) {
        *err_out = 0;
        if (h->tp) {
            MM_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
            MM_CHECK_HIP(hipStreamSynchronize(h->tp->sc));
            MM_CHECK_HIP(hipMemcpy(err_out, h->tp->err, sizeof(int), hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

int mmada_comm_set_mode(mmada_handle* h, int mode) {
    if (!h || !h->tp) return mm_fail("mmada_comm_set_mode: no comm");
    TpComm* c = h->tp;
    const int other = c->size == 1 ? 0 : (c->rank == 0 ? 1 : 0);
    if (mode == 1 && !c->peers.ctr[other]) return mm_fail("mmada_comm_set_mode: the pull transport was never connected");
    if (mode == 2 && !c->comm) return mm_fail("mmada_comm_set_mode: the RCCL transport was never connected");
    if (mode == 3 && c->mode == 0) return mm_fail("mmada_comm_set_mode: connect a transport before the no-exchange diagnostic");
    if (mode == 4 && !c->peers.ctr[other]) return mm_fail("mmada_comm_set_mode: the copy transport needs the mapped peer buffers (connect_ipc / connect_local)");
    if (mode < 1 || mode > 4) return mm_fail("mmada_comm_set_mode: mode must be 1 (pull), 2 (RCCL), 3 (diagnostic: no exchange) or 4 (copy engines)");
    if (mode == 4 && !c->stage) MM_CHECK_HIP(hipMalloc(&c->stage, c->stage_stride * c->size * 2));  // pull / RCCL users never pay for it
    c->mode = mode;
    return 0;
}

/* Ranks of the RCCL communicator this handle created (ncclCommCount), 0 when none was created. */
int mmada_comm_rccl_nranks(mmada_handle* h) {
    if (!h || !h->tp || !h->tp->comm) return 0;
    int n = 0;
    if (h->tp->nccl.CommCount && h->tp->nccl.CommCount(h->tp->comm, &n) == ncclSuccess) return n;
    return h->tp->size;  // a librccl without ncclCommCount: the size the communicator was initialised with
}

/* CU partition of the exchange (see the header comment): exchange_cus = 0 removes it; else the exchange stream is re-created on
 * `exchange_cus` CUs (a multiple of 8: the same number on every XCD) and the forward's compute kernels run on a library stream
 * masked to the remaining CUs.  Call between forwards (synchronises the device). */
int mmada_comm_set_partition(mmada_handle* h, int exchange_cus) {
    if (!h || !h->tp) return mm_fail("mmada_comm_set_partition: no comm");
    TpComm* c = h->tp;
    int dev = 0, ncu = 0;
    MM_CHECK_HIP(hipGetDevice(&dev));
    MM_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (exchange_cus < 0 || exchange_cus % 8 || exchange_cus >= ncu)
        return mm_fail("mmada_comm_set_partition: exchange_cus=%d must be a multiple of 8 below the device's %d CUs", exchange_cus, ncu);
    MM_CHECK_HIP(hipDeviceSynchronize());
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (c->s_cmp) { (void)hipStreamDestroy(c->s_cmp); c->s_cmp = nullptr; }
    if (c->sc) { (void)hipStreamDestroy(c->sc); c->sc = nullptr; }
    c->part_cus = 0;
    // whatever happens below, the exchange keeps a usable stream: the un-partitioned one (high priority, non-blocking)
    MM_CHECK_HIP(hipStreamCreateWithPriority(&c->sc, hipStreamNonBlocking, hi));
    if (exchange_cus == 0) return 0;
    const int words = (ncu + 31) / 32;
    uint32_t mx[16] = {}, mc[16] = {};
    if (words > 16) return mm_fail("mmada_comm_set_partition: %d CUs exceed the mask buffer", ncu);
    for (int i = 0; i < ncu; ++i) (i < exchange_cus ? mx : mc)[i / 32] |= 1u << (i % 32);
    // hipExtStreamCreateWithCUMask takes no flags: both masked streams are BLOCKING, default-priority streams, i.e. implicitly
    // ordered against the legacy null stream.  A caller that drives the forward on the null stream (torch's default stream on
    // ROCm) therefore serialises with them and sees no overlap; pass a non-default stream while a partition is active
    // (bench.py and the probes do).  The partition is reported (mmada_comm_partition) only once BOTH streams exist.
    hipStream_t sx = nullptr, scmp = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&sx, words, mx);
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&scmp, words, mc);
    if (e != hipSuccess) {
        if (sx) (void)hipStreamDestroy(sx);
        return mm_fail("mmada_comm_set_partition: hipExtStreamCreateWithCUMask: %s (no partition in effect)", hipGetErrorString(e));
    }
    (void)hipStreamDestroy(c->sc);
    c->sc = sx;
    c->s_cmp = scmp;
    c->part_cus = exchange_cus;
    return 0;
}
int mmada_comm_partition(mmada_handle* h) { return h && h->tp ? h->tp->part_cus : 0; }
/* The library's exchange stream and (with a partition) masked compute stream, for probes that time kernels on them. */
int mmada_comm_streams(mmada_handle* h, void** exchange_out, void** compute_out) {
    if (!h || !h->tp) return mm_fail("mmada_comm_streams: no comm");
    if (exchange_out) *exchange_out = (void*)h->tp->sc;
    if (compute_out) *compute_out = (void*)h->tp->s_cmp;
    return 0;
}

void* mmada_comm_part_ptr(mmada_handle* h) { return h && h->tp ? (void*)h->tp->part : nullptr; }

/* One exchange over every row of the resident (B, L) carve, for probes and self-tests: the caller filled the partial
 * buffer (mmada_comm_part_ptr, [B*Lp, d]) and the residual stream; afterwards x holds own rows + sum and xn (debug buffer
 * 0) the RMSNorm of every row with norm_w (device bf16 [d]).  Runs on `stream` (no second stream, no chunking). */
int mmada_comm_exchange(mmada_handle* h, const void* norm_w, void* stream) {
    if (!h || !h->tp || h->M == 0 || !norm_w) return mm_fail("mmada_comm_exchange: need a comm and a resident carve (mmada_embed)");
    const Slice sl = chunk_slice(h->M, h->tp->size, h->tp->rank, 1, 0);
    h->xn_is_layer0 = false;  // xn is about to be overwritten
    if (h->tp->mode == 1 || h->tp->mode == 4) hipLaunchKernelGGL(tp_flush_kernel, dim3(256), dim3(64), 0, (hipStream_t)stream);
    return exchange(h, sl, (const bf16_t*)norm_w, (hipStream_t)stream);
}

/* Vocabulary-parallel text step (generators/parallel_generator.py:185-217 at text_temperature == 0) after a
 * tensor-parallel forward: this rank multiplies the ln_f rows by ITS slice of ff_out.weight (vocab / tp_size columns; the
 * [B*T, vocab] logits exist nowhere), reduces each row to {max, first arg-max, fp64 sum-exp}, the tp records are exchanged
 * (16 bytes per row and rank) and combined, and the k[b] most confident masked positions are committed on every rank.
 * rows: device int32 [B*T] = b*L + text_start + t.  scratch: device, >= B*T*16 bytes (receives conf f64 / x0 i32). */
int mmada_text_select_tp(mmada_handle* h, const int32_t* rows, int B, int T, int64_t* ids, int L, int text_start,
                         const int32_t* k, void* scratch, void* stream) {
    if (!h || !h->tp || h->tp->mode == 0 || h->tp->mode == 3) return mm_fail("mmada_text_select_tp: no tensor-parallel transport connected");
    if (!h->xn_is_final || h->M == 0) return mm_fail("mmada_text_select_tp: no tensor-parallel forward resident");
    if (!rows || !ids || !k || !scratch) return mm_fail("mmada_text_select_tp: null argument");
    TpComm* c = h->tp;
    const int R = B * T;
    if (R <= 0) return 0;
    if (R > STAT_ROWS || R > h->B * h->L) return mm_fail("mmada_text_select_tp: %d rows exceed the limit", R);
    if (text_start < 0 || text_start + T > L) return mm_fail("mmada_text_select_tp: text span outside the sequence");
    hipStream_t s = (hipStream_t)stream;
    const int d = h->cfg.d_model, V = h->cfg.vocab;
    const int w = ((V + c->size - 1) / c->size + 7) / 8 * 8;
    const int v0 = min(V, c->rank * w), v1 = min(V, v0 + w);
    const size_t need = (size_t)R * w * 2;
    if (need > c->head_bytes) {  // first use (or a larger batch): not capturable, like every first call
        (void)hipFree(c->head_buf);
        c->head_buf = nullptr; c->head_bytes = 0;
        MM_CHECK_HIP(hipMalloc(&c->head_buf, need));
        c->head_bytes = need;
    }
    if (tp_head_gather(h, rows, R, s)) return 1;
    if (v1 > v0) {
        GemmArgs g{};
        g.A = h->xg; g.W = h->lm_head + (size_t)v0 * d; g.C = c->head_buf;
        g.M = R; g.N = v1 - v0; g.K = d; g.lda = d; g.ldw = d; g.ldc = w;
        if (launch_gemm(EPI_STORE, g, s)) return 1;
    }
    if (launch_text_stats_partial(c->head_buf, B, T, v1 - v0, w, v0, ids, L, text_start, h->cfg.mask_token_id, c->stats_pub, s))
        return 1;
    double* conf = (double*)scratch;
    int32_t* x0 = (int32_t*)((char*)scratch + (size_t)R * 8);
    if (c->mode == 1 || c->mode == 4) {
        if (signal_wait(c, s)) return 1;
        hipLaunchKernelGGL(tp_text_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, s, c->peers, c->size, c->rank,
                           c->stats_pub, (const TextStat*)nullptr, 0, R, conf, x0);
    } else {
        ncclResult_t r = c->nccl.AllGather(c->stats_pub, c->stats_all, (size_t)R * sizeof(TextStat), ncclUint8, c->comm, s);
        if (r != ncclSuccess) return nccl_fail(c, "ncclAllGather", r);
        hipLaunchKernelGGL(tp_text_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, s, c->peers, c->size, c->rank,
                           c->stats_pub, c->stats_all, R, R, conf, x0);
    }
    MM_CHECK_HIP(hipGetLastError());
    return launch_text_commit(scratch, B, T, ids, L, text_start, k, s);
}

int mmada_comm_destroy(mmada_handle* h) {
    if (h) tp_comm_free(h);
    return 0;
}

}  // extern "C"

int tp_gather_rows(const bf16_t* src, const int32_t* rows, int R, int L, int Lp, int d, int nflat, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, s, src, rows, R, L, Lp, d, nflat, out);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

int tp_head_gather(mmada_handle* h, const int32_t* rows, int R, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, s, h->xn, rows, R, h->L, h->Lp, h->cfg.d_model,
                       h->B * h->L, h->xg);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

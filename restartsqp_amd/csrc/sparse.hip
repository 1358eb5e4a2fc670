// sparse.hip -- HBM-bound sparse kernels of the QP path for gfx950.
//
//   * csx_stream_spmv : out[major] = sum_k val[k] * in[idx[k]]  over a compressed-major
//     matrix. On the CSC arrays of A it is SpHbMat::transposed_times (A'y, reference
//     src/SpHbMat.cpp:659-696); on the CSR copy of A, or on the symmetric H, it is
//     SpHbMat::times (A x / H x, :698-737). LDS-staged "stream" form: a workgroup streams a
//     contiguous chunk of (val, idx) with fully coalesced loads, multiplies with the
//     gathered input (the input vector is L2-resident), parks the products in LDS and
//     then reduces each short segment from LDS -- no atomics, fixed summation order
//     (entry order inside a segment, exactly the order of the reference loops).
//     Batched over blockIdx.y for independent matrices (one QP each).
//   * scatter_values : SpHbMat::setMatVal (src/SpHbMat.cpp:368-393) -- value refresh of
//     the device CSC through the permutation `order`, plus the gather that refreshes
//     the CSR copy.
//   * kkt_* : fused qpOASESInterface::get_working_set (+ the A x product) and
//     ::test_optimality (src/qpOASESInterface.cpp:498-684, 835-895).
#include "rsqp_sparse.h"

namespace {

constexpr int SPMV_NT = 256;
constexpr int SPMV_CHUNK = 2048;  // entries staged per workgroup: 16 KiB of LDS
constexpr int SPMV_ITERS = SPMV_CHUNK / (2 * SPMV_NT);

// blkinfo[b] = {first major, end major, first entry, end entry} of block b (one 16-byte load
// instead of the dependent chain blk -> ptr). With xcd_map the 1-D grid is decoded so that
// all blocks of one batch member carry the same (blockIdx.x % 8): workgroups are dealt
// round-robin over the 8 XCDs, so a member's gathered input vector is pulled into ONE L2
// instead of eight (placement only affects speed, never results).
// EPI: the diagonal range-space path's step direction rides behind A'dy_C (RsqpSpmvDualDx, rsqp_sparse.h): the thread that has
// formed entry v of the product applies what was the kernel k_dual_dx to it -- one launch less per working-set change
template <bool EPI>
__device__ __forceinline__ void spmv_out(double *__restrict__ out, int r, double s, const RsqpSpmvDualDx &e) {
    out[r] = s;
    if constexpr (EPI) {
        double d = e.dx[r];
        if (e.Sb[r] == 0) { d = e.hinv[r] * (s - (e.gN[r] - e.g[r])); e.dx[r] = d; }
        e.Hdx[r] = (e.Hval[r] + e.hreg) * d;
    }
}
template <bool EPI>
__global__ void __launch_bounds__(SPMV_NT)
csx_stream_spmv(const int4 *__restrict__ blkinfo, int nblk, int nbatch, int xcd_map,
                const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
                const double *__restrict__ in, double *__restrict__ out, long long ptr_stride,
                long long nnz_stride, long long in_stride, long long out_stride, RsqpSpmvDualDx epi) {
    __shared__ __attribute__((aligned(16))) double prod[SPMV_CHUNK + 2];
    int m, b;
    if (xcd_map) {
        const int L = blockIdx.x, j = L >> 3;
        m = (j / nblk) * 8 + (L & 7);
        b = j % nblk;
        if (m >= nbatch) return;
    } else {
        m = blockIdx.x / nblk;
        b = blockIdx.x % nblk;
    }
    ptr += m * ptr_stride; idx += m * nnz_stride; val += m * nnz_stride;
    in += m * in_stride; out += m * out_stride;
    const int4 bi = blkinfo[b];
    const int r0 = bi.x, r1 = bi.y, k0 = bi.z, k1 = bi.w;
    const int tid = threadIdx.x;
    if (k1 - k0 <= SPMV_CHUNK - 2) {
        // segment bounds of "my" major: issued now, consumed after the barrier
        const int myr = r0 + tid;
        int pa = 0, pb = 0;
        if (myr < r1) { pa = ptr[myr]; pb = ptr[myr + 1]; }
        // stream (val, idx) in 16-byte / 8-byte pieces from an even entry index; all loads are
        // issued before the first gather so that every lane keeps 2*ITERS requests in flight
        const int ka = k0 & ~1, klast = (k1 - 1) & ~1;
        double2 vv[SPMV_ITERS];
        int2 ii[SPMV_ITERS];
#pragma unroll
        for (int it = 0; it < SPMV_ITERS; it++) {
            int k = ka + 2 * tid + it * 2 * SPMV_NT;
            k = k < klast ? k : klast;  // clamp: the load is always inside this block's range
            vv[it] = *reinterpret_cast<const double2 *>(val + k);
            ii[it] = *reinterpret_cast<const int2 *>(idx + k);
        }
#pragma unroll
        for (int it = 0; it < SPMV_ITERS; it++) {
            const int k = ka + 2 * tid + it * 2 * SPMV_NT;
            if (k < k1) {
                const bool v0 = k >= k0, v1 = k + 1 < k1;
                double2 p;
                p.x = v0 ? vv[it].x * in[ii[it].x] : 0.0;
                p.y = v1 ? vv[it].y * in[ii[it].y] : 0.0;
                *reinterpret_cast<double2 *>(prod + (k - ka)) = p;
            }
        }
        __syncthreads();
        if (myr < r1) {
            double s = 0.0;
            for (int k = pa - ka; k < pb - ka; k++) s += prod[k];
            spmv_out<EPI>(out, myr, s, epi);
        }
        for (int r = myr + SPMV_NT; r < r1; r += SPMV_NT) {
            double s = 0.0;
            const int a = ptr[r] - ka, e = ptr[r + 1] - ka;
            for (int k = a; k < e; k++) s += prod[k];
            spmv_out<EPI>(out, r, s, epi);
        }
    } else {
        // a single long segment (the block builder never mixes it with others):
        // every lane accumulates a strided slice, fixed-order tree afterwards
        double s = 0.0;
        for (int k = k0 + tid; k < k1; k += SPMV_NT) s += val[k] * in[idx[k]];
        prod[tid] = s;
        __syncthreads();
        for (int o = SPMV_NT / 2; o > 0; o >>= 1) {
            if (tid < o) prod[tid] += prod[tid + o];
            __syncthreads();
        }
        if (tid == 0) spmv_out<EPI>(out, r0, prod[0], epi);
    }
}

// ---------------------------------------------------------------------------------
// Batched product with the INPUT VECTOR RESIDENT IN LDS (MI355X: 160 KiB per CU holds the
// whole 20 000-entry fp64 vector of the n=10k x m=20k configuration). One 1024-thread
// workgroup per (batch member, slice of majors): it loads the member's vector once with
// 16-byte loads, then streams its (val, idx) range from HBM while every gather is a
// ds_read_b64 instead of a 64-byte L2 sector request. A sub-wave of G lanes owns one major
// at a time; consecutive sub-waves own consecutive majors, so a wave-instruction touches a
// contiguous run of entries. No barrier after the vector load, no atomics; fixed-shape
// reduction (lane-strided partial sums + xor tree) => run-to-run deterministic.
// ---------------------------------------------------------------------------------
constexpr int LV_NT = 1024;

template <int G, int U>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_spmv(int nminor, int nslices, const int *__restrict__ slice, const int *__restrict__ ptr,
                const int *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ in,
                double *__restrict__ out, long long ptr_stride, long long nnz_stride, long long in_stride,
                long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x / nslices, sl = blockIdx.x % nslices;
    ptr += m * ptr_stride; idx += m * nnz_stride; val += m * nnz_stride;
    in += m * in_stride; out += m * out_stride;
    const int tid = threadIdx.x;
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    if (tid == 0 && (nminor & 1)) xs[nminor - 1] = in[nminor - 1];
    __syncthreads();
    constexpr int NG = LV_NT / G;
    const int gid = tid / G, gl = tid % G;
    const int c1 = slice[sl + 1];
    for (int c = slice[sl] + gid; c < c1; c += NG) {
        const int a = ptr[c], b = ptr[c + 1];
        double s = 0.0;
        // U predicated steps with clamped addresses: all 2U loads are issued back to back
        double v[U];
        int j[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int k = a + gl + u * G;
            k = k < b ? k : (b > a ? b - 1 : a);
            v[u] = val[k];
            j[u] = idx[k];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (a + gl + u * G < b) s += v[u] * xs[j[u]];
        for (int k = a + gl + U * G; k < b; k += G) s += val[k] * xs[idx[k]];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (gl == 0) out[c] = s;
    }
}

// software-pipelined form of csx_ldsvec_spmv: the segment bounds are requested two majors
// ahead and the (val, idx) loads one major ahead of the LDS gathers that consume them, so a
// lane always has 2U + 2 global loads in flight (loads return in order: the wait for the
// current major's data does not drain the next major's requests).
template <int G, int U, bool NTL>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_spmv_pipe(int nminor, int nslices, const int *__restrict__ slice, const int *__restrict__ ptr,
                     const int *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ in,
                     double *__restrict__ out, long long ptr_stride, long long nnz_stride, long long in_stride,
                     long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x / nslices, sl = blockIdx.x % nslices;
    ptr += m * ptr_stride; idx += m * nnz_stride; val += m * nnz_stride;
    in += m * in_stride; out += m * out_stride;
    const int tid = threadIdx.x;
    constexpr int NG = LV_NT / G;
    const int gid = tid / G, gl = tid % G;
    const int c1 = slice[sl + 1];
    int c = slice[sl] + gid;
    // bounds of the first two majors of this sub-wave (clamped: always a valid address)
    const int cl = c1 - 1;
    int a0 = ptr[c < c1 ? c : cl], b0 = ptr[(c < c1 ? c : cl) + 1];
    int a1 = ptr[c + NG < c1 ? c + NG : cl], b1 = ptr[(c + NG < c1 ? c + NG : cl) + 1];
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    if (tid == 0 && (nminor & 1)) xs[nminor - 1] = in[nminor - 1];
    double v[U];
    int j[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        int k = a0 + gl + u * G;
        k = k < b0 ? k : (b0 > a0 ? b0 - 1 : a0);
        v[u] = NTL ? __builtin_nontemporal_load(val + k) : val[k];
        j[u] = NTL ? __builtin_nontemporal_load(idx + k) : idx[k];
    }
    __syncthreads();
    for (; c < c1; c += NG) {
        // bounds two majors ahead
        const int c2 = c + 2 * NG < c1 ? c + 2 * NG : cl;
        const int a2 = ptr[c2], b2 = ptr[c2 + 1];
        // data one major ahead
        double vn[U];
        int jn[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int k = a1 + gl + u * G;
            k = k < b1 ? k : (b1 > a1 ? b1 - 1 : a1);
            vn[u] = NTL ? __builtin_nontemporal_load(val + k) : val[k];
            jn[u] = NTL ? __builtin_nontemporal_load(idx + k) : idx[k];
        }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (a0 + gl + u * G < b0) s += v[u] * xs[j[u]];
        for (int k = a0 + gl + U * G; k < b0; k += G) s += val[k] * xs[idx[k]];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (gl == 0) out[c] = s;
#pragma unroll
        for (int u = 0; u < U; u++) { v[u] = vn[u]; j[u] = jn[u]; }
        a0 = a1; b0 = b1; a1 = a2; b1 = b2;
    }
}

// two consecutive entries per lane (16-byte value load) and an index type template: when
// the gathered vector has fewer than 65 536 entries the plan keeps a 16-bit copy of the index
// array, which cuts the streamed bytes per entry from 12 to 10.
template <class IDX> struct Idx2;
template <> struct Idx2<int> { typedef int2 T; };
template <> struct Idx2<unsigned short> { typedef ushort2 T; };

template <int G, int U, class IDX>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_spmv_pipe2(int nminor, int nslices, const int *__restrict__ slice, const int *__restrict__ ptr,
                      const IDX *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ in,
                      double *__restrict__ out, long long ptr_stride, long long nnz_stride, long long in_stride,
                      long long out_stride) {
    typedef typename Idx2<IDX>::T I2;
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x / nslices, sl = blockIdx.x % nslices;
    ptr += m * ptr_stride; idx += m * nnz_stride; val += m * nnz_stride;
    in += m * in_stride; out += m * out_stride;
    const int tid = threadIdx.x;
    constexpr int NG = LV_NT / G;
    const int gid = tid / G, gl = tid % G;
    const int c1 = slice[sl + 1];
    int c = slice[sl] + gid;
    const int cl = c1 - 1;
    int a0 = ptr[c < c1 ? c : cl], b0 = ptr[(c < c1 ? c : cl) + 1];
    int a1 = ptr[c + NG < c1 ? c + NG : cl], b1 = ptr[(c + NG < c1 ? c + NG : cl) + 1];
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    if (tid == 0 && (nminor & 1)) xs[nminor - 1] = in[nminor - 1];
    double2 v[U];
    I2 j[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        int k = a0 + 2 * (gl + u * G);
        const int kc = b0 - 2 > a0 ? b0 - 2 : a0;
        k = k < kc ? k : kc;
        __builtin_memcpy(&v[u], val + k, 16);
        __builtin_memcpy(&j[u], idx + k, sizeof(I2));
    }
    __syncthreads();
    for (; c < c1; c += NG) {
        const int c2 = c + 2 * NG < c1 ? c + 2 * NG : cl;
        const int a2 = ptr[c2], b2 = ptr[c2 + 1];
        double2 vn[U];
        I2 jn[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int k = a1 + 2 * (gl + u * G);
            const int kc = b1 - 2 > a1 ? b1 - 2 : a1;
            k = k < kc ? k : kc;
            __builtin_memcpy(&vn[u], val + k, 16);
            __builtin_memcpy(&jn[u], idx + k, sizeof(I2));
        }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = a0 + 2 * (gl + u * G);
            const int kc = b0 - 2 > a0 ? b0 - 2 : a0;
            // a clamped lane re-reads the last pair: use it only when k itself is in range
            if (k <= kc && k < b0) {
                s += v[u].x * xs[j[u].x];
                if (k + 1 < b0) s += v[u].y * xs[j[u].y];
            } else if (k < b0) {   // k == b0 - 1 > kc: single last entry sits in .y of the clamped pair
                s += v[u].y * xs[j[u].y];
            }
        }
        for (int k = a0 + gl + 2 * U * G; k < b0; k += G) s += val[k] * xs[idx[k]];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (gl == 0) out[c] = s;
#pragma unroll
        for (int u = 0; u < U; u++) { v[u] = vn[u]; j[u] = jn[u]; }
        a0 = a1; b0 = b1; a1 = a2; b1 = b2;
    }
}

// ---------------------------------------------------------------------------------
// Entry-parallel batched product with the input vector resident in LDS (variant 40). The sub-wave-per-major kernels
// above read every major as its own little segment: a wave instruction touches 16 pieces of 64 bytes and the value
// stream runs at 4.4 TB/s, while the pure stream of this shape runs at 6.2 TB/s (tools/spmv_bound_check.py 50).
// Here the plan cuts the entry stream into CHUNKS of whole majors with at most 512 entries and 128 majors; a wave owns
// a chunk (no sum ever crosses a wave), a lane owns 8 CONSECUTIVE entries of it:
//   * loads: 16 bytes of indices per lane = 1 KB contiguous per wave; 4 x 16 bytes of values per lane (the four
//     instructions of a wave sweep the same 4 KB, sector by sector);
//   * "this entry starts a major" is one bit per entry (a byte per lane, 64 bytes per chunk); the pointer array is
//     never read -- the ordinal of a major is a prefix count of those bits;
//   * a lane forms all prefix and suffix sums of its 8 products (two FMA chains), from which the piece before its first
//     start and the piece from its last start on are picked; majors wholly inside a lane are summed in a loop only lanes
//     with two starts enter; the piece running in from the previous lanes and the piece running out meet in ONE
//     segmented scan over the 64 lanes (DPP row shifts and row broadcasts: no LDS traffic), after which the lanes
//     holding the end of a major emit its sum (straight to memory, or through a 1 KB LDS window of the wave that leaves
//     as one run of consecutive stores at the top of the next trip -- STAGE);
//   * behind the last entry of a chunk that is not full the plan sets one more start bit: what a lane loaded past the
//     chunk forms a dummy major that is never emitted (no validity masks in the inner code).
// Fixed-shape tree per major => run-to-run deterministic; not the entry order of the reference loop (tolerance test).
// ---------------------------------------------------------------------------------
struct SegPlanView {
    const int4 *chunks;       // per chunk: {first entry, ordinal of its first major in nzlist, entries, majors}
    const int *nzlist;        // the non-empty majors, in order
    const int *empties;       // the empty majors (their output is zero)
    int nchunks, nempty;
};

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double x) {      // lanes without a source (or in a masked row) receive 0.0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i32(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xf, false); }
// one round of the inclusive segmented scan: (pv, pf) is the aggregate of the lanes before, where there is one
// DIRECT: the pattern has no empty major, so the ordinal of a major IS its index (no look-up in the store path)
// STAGE: the sums of a chunk (at most 128 majors, consecutive ordinals) are collected in a 1 KB LDS window of the wave and
// leave as ONE run of consecutive 8-byte stores instead of one scattered store per lane and store site
template <bool DIRECT, bool STAGE>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_segscan(int nminor, SegPlanView sp, const unsigned char *__restrict__ sbits, const unsigned short *__restrict__ idx,
                   const double *__restrict__ val, const double *__restrict__ in, double *__restrict__ out,
                   long long nnz_stride, long long in_stride, long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x;
    idx += m * nnz_stride; val += m * nnz_stride; in += m * in_stride; out += m * out_stride;
    sbits += (long long)m * sp.nchunks * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = LV_NT / 64;
    double *const stg = xs + ((nminor + 1) & ~1) + wave * 128;
    struct Ld { double2 v[4]; ushort4 ja, jb; unsigned char b; };
    // lane l holds the entries [8 l, 8 l + 8) of the chunk; what lies past the chunk (the next chunk's entries, or
    // the padding behind the last member) is loaded too and ends up in a dummy major (see the start bits below)
    auto issue = [&](const int4 &c, int kk, Ld &d) {
        const long long e = (long long)c.x + 8 * lane;
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_memcpy(&d.v[q], val + e + 2 * q, 16);
        __builtin_memcpy(&d.ja, idx + e, 8);
        __builtin_memcpy(&d.jb, idx + e + 4, 8);
        d.b = sbits[(kk < sp.nchunks ? kk : sp.nchunks - 1) * 64 + lane];      // unconditional (clamped): a branch around a
                                                                                // load makes the compiler drain the prefetch
    };
    // one chunk: the lane's 8 products, the majors inside the lane, the segmented scan over the lanes, the stores
    auto reduce_chunk = [&](const Ld &cur, const int4 &ch) __attribute__((always_inline)) {
        auto put = [&](int o, double t) {
            if (STAGE) stg[o - ch.y] = t;
            else out[DIRECT ? o : sp.nzlist[o]] = t;
        };
        const int n = ch.z;
        const unsigned bits = cur.b;         // bit i: entry 8 lane + i starts a major; behind the last entry of a chunk that
                                             // is not full the plan sets ONE more bit, so what a lane loaded past the chunk
                                             // forms a dummy major that is never stored (no validity masks needed)
        const int nflag = __popc(bits);
        // starts in the lanes before this one -> ordinal of the first major that starts in this lane
        int sc = nflag;
        sc += dpp_i32<0x111, 0xf>(sc); sc += dpp_i32<0x112, 0xf>(sc); sc += dpp_i32<0x114, 0xf>(sc); sc += dpp_i32<0x118, 0xf>(sc);
        sc += dpp_i32<0x142, 0xa>(sc); sc += dpp_i32<0x143, 0xc>(sc);
        const int ord0 = ch.y + sc - nflag;
        const unsigned short jj[8] = {cur.ja.x, cur.ja.y, cur.ja.z, cur.ja.w, cur.jb.x, cur.jb.y, cur.jb.z, cur.jb.w};
        double av[8], xv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { av[i] = (i & 1) ? cur.v[i >> 1].y : cur.v[i >> 1].x; xv[i] = xs[jj[i]]; }
        // all prefix sums S[i] = p_0 + .. + p_i and all suffix sums T[i] = p_i + .. + p_7 of the lane's products, one
        // FMA each: the part before the first start is S[first - 1], the part from the last start on is T[last]
        double S[8], T[8];
        S[0] = av[0] * xv[0];
#pragma unroll
        for (int i = 1; i < 8; i++) S[i] = fma(av[i], xv[i], S[i - 1]);
        T[7] = av[7] * xv[7];
#pragma unroll
        for (int i = 6; i >= 0; i--) T[i] = fma(av[i], xv[i], T[i + 1]);
        const int seen = bits != 0;
        const int j1 = seen ? __ffs(bits) - 1 : 8, jl = seen ? 31 - __clz(bits) : 0;
        auto pick8 = [](double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7, int t) {
            // element t of eight registers as a select tree (registers cannot be indexed)
            const double b0 = (t & 1) ? a1 : a0, b1 = (t & 1) ? a3 : a2, b2 = (t & 1) ? a5 : a4, b3 = (t & 1) ? a7 : a6;
            const double c0 = (t & 2) ? b1 : b0, c1 = (t & 2) ? b3 : b2;
            return (t & 4) ? c1 : c0;
        };
        const double head = pick8(S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], (j1 + 7) & 7);          // S[j1 - 1] (unused when j1 == 0); no start at all: S[7], the whole lane
        const double acc = seen ? pick8(T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7], jl) : head;       // the piece running out of the lane
        // majors that start AND end inside the lane (short ones; also the last one of a chunk that is not full): summed and
        // stored here, one per trip of a loop that only lanes with two or more starts enter
        {
            unsigned bb = bits;
            int a = j1, kk = 0;
            bb &= bb - 1;
            while (bb) {
                const int b = __ffs(bb) - 1;
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < 8; i++) t = (i >= a && i < b) ? fma(av[i], xv[i], t) : t;
                if (8 * lane + a < n) put(ord0 + kk, t);
                a = b; kk++;
                bb &= bb - 1;
            }
        }
        // segmented inclusive scan over the lanes of (piece running out of the lane, any start in the lane)
        double sv = acc;
        int sf = seen;
#define SEG_ROUND(CTRL, ROWMASK)                                         \
        do {                                                             \
            const double pv_ = dpp_f64<CTRL, ROWMASK>(sv);               \
            const int pf_ = dpp_i32<CTRL, ROWMASK>(sf);                  \
            if (!sf) sv += pv_;                                          \
            sf |= pf_;                                                   \
        } while (0)
        SEG_ROUND(0x111, 0xf); SEG_ROUND(0x112, 0xf); SEG_ROUND(0x114, 0xf); SEG_ROUND(0x118, 0xf);   // row_shr 1, 2, 4, 8
        SEG_ROUND(0x142, 0xa);      // row_bcast15 into rows 1 and 3
        SEG_ROUND(0x143, 0xc);      // row_bcast31 into rows 2 and 3
#undef SEG_ROUND
        double ex = __shfl_up(sv, 1);        // the sum running into this lane
        if (lane == 0) ex = 0.0;
        int nextstart = __shfl_down((int)(bits & 1u), 1);
        if (lane == 63) nextstart = 1;
        const int lastlane = (n - 1) >> 3;
        // the piece running out of this lane ends a major when the next lane begins with a start; in the last lane of the
        // chunk only when the chunk fills it (otherwise that piece is the dummy major behind the chunk)
        const bool ends_here = lane == lastlane ? (n & 7) == 0 : (lane < lastlane && nextstart);
        if (lane <= lastlane) {
            if (seen) {
                if (j1 > 0) put(ord0 - 1, ex + head);                  // runs in from before, ends inside this lane
                if (ends_here) put(ord0 + nflag - 1, acc);
            } else if (ends_here) {
                put(ord0 - 1, ex + acc);
            }
        }
        if (STAGE) __builtin_amdgcn_wave_barrier();  // LDS operations of a wave execute in order; keep the compiler from moving them
    };
    // STAGE: the sums of the chunk reduced in the PREVIOUS trip leave the LDS window at the top of the next trip, BEFORE
    // that trip's loads are issued. Loads and stores share one in-order counter (vmcnt) and the compiler waits for a store
    // before it reuses the store's operand registers: a store issued after the prefetch loads makes that wait drain the
    // prefetch; issued before them, the wait leaves the newest chunk in flight.
    auto flush = [&](int base, int count) __attribute__((always_inline)) {
        if (!STAGE) return;
        for (int r = lane; r < count; r += 64) out[DIRECT ? base + r : sp.nzlist[base + r]] = stg[r];
        __builtin_amdgcn_wave_barrier();
    };
    int k = wave;
    auto chunk_at = [&](int kk) {           // past the end: the last chunk's addresses (valid loads), never reduced
        return sp.chunks[kk < sp.nchunks ? kk : sp.nchunks - 1];
    };
    // TWO chunks (10 KB per wave, 160 KB per CU) stay in flight while one is reduced: with one, the 16 waves of a CU
    // cover only ~80 KB of the latency x bandwidth product. Three buffers change roles in a loop unrolled by three --
    // copying "next" into "current" would wait for the loads just issued and undo the prefetch
    // (the descriptor of a chunk is fetched one trip before its loads are issued, so no trip waits for a scalar load)
    int4 c0 = chunk_at(k), c1 = chunk_at(k + NWV), c2, cn = chunk_at(k + 2 * NWV);
    Ld b0, b1, b2;
    issue(c0, k, b0);                        // the first two chunks' loads go out before the vector is staged
    issue(c1, k + NWV, b1);
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    if (tid == 0 && (nminor & 1)) xs[nminor - 1] = in[nminor - 1];
    for (int i = tid; i < sp.nempty; i += LV_NT) out[sp.empties[i]] = 0.0;
    __syncthreads();
    int pbase = 0, pcount = 0;
#define SEG_TRIP(BN, CN, BC, CC)                                                                  \
        flush(pbase, pcount);                                                                     \
        CN = cn; cn = chunk_at(k + 3 * NWV);                                                      \
        issue(CN, k + 2 * NWV, BN);                                                               \
        pcount = 0;                                                                               \
        if (k < sp.nchunks) { reduce_chunk(BC, CC); pbase = CC.y; pcount = CC.w; }                \
        k += NWV;
    while (k < sp.nchunks) {        // no exit inside the body: the trips past the end load the last chunk again and skip the reduction
        SEG_TRIP(b2, c2, b0, c0)
        SEG_TRIP(b0, c0, b1, c1)
        SEG_TRIP(b1, c1, b2, c2)
    }
#undef SEG_TRIP
    flush(pbase, pcount);
}

#ifdef RSQP_SPMV_EXPERIMENT
// TIMING EXPERIMENT ONLY (never selected by the plan): upper bound of an entry-parallel design -- every wave streams
// contiguous 128-entry steps fully coalesced (16-byte value + 4-byte index loads), gathers from the LDS vector and
// just accumulates; the result is meaningless (no segmented reduction)
template <int U>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_streambound(int nminor, int nnz, const unsigned short *__restrict__ idx, const double *__restrict__ val,
                       const double *__restrict__ in, double *__restrict__ out, long long nnz_stride, long long in_stride,
                       long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x;
    idx += m * nnz_stride; val += m * nnz_stride; in += m * in_stride; out += m * out_stride;
    const int tid = threadIdx.x;
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    __syncthreads();
    double s = 0.0;
    const int step = 2 * LV_NT * U;
    for (int base = 0; base < nnz; base += step) {
        double2 v[U];
        ushort2 j[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int e = base + u * 2 * LV_NT + 2 * tid;
            e = e + 1 < nnz ? e : nnz - 2;
            __builtin_memcpy(&v[u], val + e, 16);
            __builtin_memcpy(&j[u], idx + e, 4);
        }
#pragma unroll
        for (int u = 0; u < U; u++) s += v[u].x * xs[j[u].x] + v[u].y * xs[j[u].y];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) out[tid >> 6] = s;
}
// the load pattern of csx_ldsvec_segscan alone: 8 consecutive entries per lane (4 x 16 B of values at a 64-B lane
// stride, 16 B of indices), chunks of 512 entries per wave, next chunk in flight; no reduction logic
// WORK: dummy work between the issue of the next chunk's loads and the use of this one's, to see what the stream
// tolerates: 0 none; 1: 192 dependent f64 FMAs; 2: 24 dependent DPP row shifts + adds (f64); 3: 12 ds_bpermute round
// trips; 4: 1 + 2 + 3
template <int WORK>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_lanegroup_bound(int nminor, int nnz, const unsigned short *__restrict__ idx, const double *__restrict__ val,
                           const double *__restrict__ in, double *__restrict__ out, long long nnz_stride, long long in_stride,
                           long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x;
    idx += m * nnz_stride; val += m * nnz_stride; in += m * in_stride; out += m * out_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    __syncthreads();
    double s = 0.0;
    const int nch = nnz / 512;
    double2 v[4], vn[4];
    ushort4 ja, jb, jan, jbn;
    auto issue = [&](int k, double2 *vv, ushort4 &a, ushort4 &b) {
        const long long e = (long long)(k < nch ? k : 0) * 512 + 8 * lane + 1;    // + 1: misaligned like a real chunk start
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_memcpy(&vv[q], val + e + 2 * q, 16);
        __builtin_memcpy(&a, idx + e, 8);
        __builtin_memcpy(&b, idx + e + 4, 8);
    };
    issue(wave, v, ja, jb);
    for (int k = wave; k < nch; k += LV_NT / 64) {
        issue(k + LV_NT / 64, vn, jan, jbn);
        s += v[0].x * xs[ja.x] + v[0].y * xs[ja.y] + v[1].x * xs[ja.z] + v[1].y * xs[ja.w] +
             v[2].x * xs[jb.x] + v[2].y * xs[jb.y] + v[3].x * xs[jb.z] + v[3].y * xs[jb.w];
        if (WORK == 1 || WORK == 4) {
#pragma unroll
            for (int q = 0; q < 192; q++) s = fma(s, 1.0000001, 1e-9);
        }
        if (WORK == 2 || WORK == 4) {
#pragma unroll
            for (int q = 0; q < 24; q++) s += dpp_f64<0x111, 0xf>(s);
        }
        if (WORK == 3 || WORK == 4) {
#pragma unroll
            for (int q = 0; q < 12; q++) s += __shfl_up(s, 1);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = vn[q];
        ja = jan; jb = jbn;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wave] = s;
}

// the same with TWO chunks in flight (three buffers, unrolled by three, no exit inside the loop)
template <int WORK>
__global__ void __launch_bounds__(LV_NT)
csx_ldsvec_lanegroup_bound2(int nminor, int nnz, const unsigned short *__restrict__ idx, const double *__restrict__ val,
                            const double *__restrict__ in, double *__restrict__ out, long long nnz_stride, long long in_stride,
                            long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int m = blockIdx.x;
    idx += m * nnz_stride; val += m * nnz_stride; in += m * in_stride; out += m * out_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = 2 * tid; i + 1 < nminor; i += 2 * LV_NT)
        *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(in + i);
    __syncthreads();
    double s = 0.0;
    const int nch = nnz / 512;
    constexpr int NWV = LV_NT / 64;
    struct B { double2 v[4]; ushort4 ja, jb; };
    auto issue = [&](int k, B &b) {
        const long long e = (long long)(k < nch ? k : 0) * 512 + 8 * lane + 1;
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_memcpy(&b.v[q], val + e + 2 * q, 16);
        __builtin_memcpy(&b.ja, idx + e, 8);
        __builtin_memcpy(&b.jb, idx + e + 4, 8);
    };
    auto work = [&](const B &b) __attribute__((always_inline)) {
        s += b.v[0].x * xs[b.ja.x] + b.v[0].y * xs[b.ja.y] + b.v[1].x * xs[b.ja.z] + b.v[1].y * xs[b.ja.w] +
             b.v[2].x * xs[b.jb.x] + b.v[2].y * xs[b.jb.y] + b.v[3].x * xs[b.jb.z] + b.v[3].y * xs[b.jb.w];
        if (WORK == 1 || WORK == 4) {
#pragma unroll
            for (int q = 0; q < 192; q++) s = fma(s, 1.0000001, 1e-9);
        }
        if (WORK == 2 || WORK == 4) {
#pragma unroll
            for (int q = 0; q < 24; q++) s += dpp_f64<0x111, 0xf>(s);
        }
        if (WORK == 3 || WORK == 4) {
#pragma unroll
            for (int q = 0; q < 12; q++) s += __shfl_up(s, 1);
        }
    };
    B b0, b1, b2;
    int k = wave;
    issue(k, b0); issue(k + NWV, b1);
    while (k < nch) {
        issue(k + 2 * NWV, b2); if (k < nch) work(b0); k += NWV;
        issue(k + 2 * NWV, b0); if (k < nch) work(b1); k += NWV;
        issue(k + 2 * NWV, b1); if (k < nch) work(b2); k += NWV;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wave] = s;
}
#endif

__global__ void scatter_values(int n, const int *__restrict__ order, const int *__restrict__ tmap,
                               const double *__restrict__ tv, double *__restrict__ val) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) val[order[i]] = tv[tmap ? tmap[i] : i];
}

// setMatVal of a matrix that also keeps a CSR copy, ONE launch (SpHbMat.cpp:368-380): triplet value i goes to its CSC slot
// order[i] and to its CSR slot rorder[i] = the CSR position of that entry (composed on the host at structure time); the
// identity entries behind the first n are never rewritten, exactly as in the reference. 8 + 4 + 4 B read, 16 B written per
// entry, against 20 B + 20 B for the scatter followed by the gather of the whole CSR copy.
__global__ void scatter_values_csc_csr(int n, const int *__restrict__ order, const int *__restrict__ rorder,
                                       const double *__restrict__ tv, double *__restrict__ val, double *__restrict__ rval) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double v = tv[i]; val[order[i]] = v; rval[rorder[i]] = v; }
}

// dense column-major copy of a CSC matrix (the target is zero-filled beforehand)
__global__ void densify_csc(int ncol, long long ld, const int *__restrict__ jc, const int *__restrict__ ir,
                            const double *__restrict__ val, double *__restrict__ dense) {
    const int c = blockIdx.x;
    if (c >= ncol) return;
    for (int k = jc[c] + threadIdx.x; k < jc[c + 1]; k += blockDim.x) dense[c * ld + ir[k]] = val[k];
}

__global__ void gather_values(int n, const int *__restrict__ perm, const double *__restrict__ src,
                              double *__restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

// ---------------------------------------------------------------------------------
// KKT certificate. One workgroup of 256 threads per QP; products come from Ax / ATy / Hx
// computed beforehand (large QPs: csx_stream_spmv) or inside (small QPs, batched).
// Sums use a fixed tree => run-to-run deterministic.
// ---------------------------------------------------------------------------------
constexpr int KKT_NT = 256;

__device__ inline double block_sum_256(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ void kkt_body(const RsqpKktArgs &a, double *sh) {
    const int q = blockIdx.x;
    const int nV = a.nV ? a.nV[q] : a.nV1, nC = a.nC ? a.nC[q] : a.nC1;
    const long long oV = a.offV ? a.offV[q] : 0, oC = a.offC ? a.offC[q] : 0;
    const double *x = a.x + oV, *y = a.y + oV + oC, *g = a.g + oV, *lb = a.lb + oV, *ub = a.ub + oV;
    const double *lbA = a.lbA + oC, *ubA = a.ubA + oC, *Ax = a.Ax + oC, *ATy = a.ATy + oV, *Hx = a.Hx + oV;
    const int *wsb = a.ws_b + oV, *wsc = a.ws_c + oC;
    int *Wb = a.W_b + oV, *Wc = a.W_c + oC;
    double primal = 0.0, dual = 0.0, compl_ = 0.0, stat = 0.0;
    int bad = 0;
    for (int v = threadIdx.x; v < nV; v += KKT_NT) {
        double xv = x[v], l = fmax(lb[v], -RSQP_K_INFTY), u = fmin(ub[v], RSQP_K_INFTY), yv = y[v];
        int W = map_bound(wsb[v], xv, l, u);
        Wb[v] = W;
        primal += fmax(0.0, l - xv) + -fmin(0.0, u - xv);            // :518-521
        kkt_terms(W, yv, xv, l, u, dual, compl_, bad);
        stat += fabs(ATy[v] + yv - g[v] - Hx[v]);                       // :595-604
    }
    for (int i = threadIdx.x; i < nC; i += KKT_NT) {
        double ax = Ax[i], l = fmax(lbA[i], -RSQP_K_INFTY), u = fmin(ubA[i], RSQP_K_INFTY), yv = y[nV + i];
        int W = map_constr(wsc[i], ax, l, u);
        Wc[i] = W;
        primal += fmax(0.0, l - ax) + -fmin(0.0, u - ax);            // :524-527
        kkt_terms(W, yv, ax, l, u, dual, compl_, bad);
    }
    if (a.done_flag) __threadfence_system();      // W_b / W_c (host-mapped for a single QP) visible before the flag below
    primal = block_sum_256(primal, sh);
    dual = block_sum_256(dual, sh);
    compl_ = block_sum_256(compl_, sh);
    stat = block_sum_256(stat, sh);
    double fb = block_sum_256((double)bad, sh);
    if (threadIdx.x == 0) {
        double *o = a.out + 6LL * q;
        o[0] = primal; o[1] = dual; o[2] = compl_; o[3] = stat;
        o[4] = compl_ + stat + dual + primal;   // :664-665
        o[5] = fb;
        if (a.done_flag && q == 0) { __threadfence_system(); *reinterpret_cast<volatile int *>(a.done_flag) = a.done_val; }
    }
}
__global__ void __launch_bounds__(KKT_NT)
kkt_kernel(RsqpKktArgs a) {
    __shared__ double sh[4];
    kkt_body(a, sh);
}

// products for a batch of SMALL problems: one workgroup per problem, lane per row/column
__device__ __forceinline__ void small_products_body(const QPDesc *desc, const int *Ajc, const int *Air, const double *Aval,
                      const int *Arp, const int *Aci, const double *Arv, const int *Hjc, const int *Hir,
                      const double *Hval, const double *x, const double *y, double *Ax, double *ATy,
                      double *Hx) {
    const QPDesc d = desc[blockIdx.x];
    const double *xq = x + d.offV, *yc = y + d.offV + d.offC + d.nV;
    for (int r = threadIdx.x; r < d.nC; r += KKT_NT) {
        const int *rp = Arp + d.offArp;
        double s = 0.0;
        for (int k = rp[r]; k < rp[r + 1]; k++) s += Arv[d.offAnz + k] * xq[Aci[d.offAnz + k]];
        Ax[d.offC + r] = s;
    }
    for (int c = threadIdx.x; c < d.nV; c += KKT_NT) {
        const int *jc = Ajc + d.offAjc;
        double s = 0.0;
        for (int k = jc[c]; k < jc[c + 1]; k++) s += Aval[d.offAnz + k] * yc[Air[d.offAnz + k]];
        ATy[d.offV + c] = s;
        double h = 0.0;
        if (d.haveH) {
            const int *hj = Hjc + d.offHjc;
            for (int k = hj[c]; k < hj[c + 1]; k++) h += Hval[d.offHnz + k] * xq[Hir[d.offHnz + k]];
        }
        Hx[d.offV + c] = h;
    }
}
// the three products and the certificate of a SMALL problem in one launch: the workgroup that
// wrote Ax / A'y / Hx reads them back after a barrier
__global__ void __launch_bounds__(KKT_NT)
small_certificate_kernel(const QPDesc *desc, const int *Ajc, const int *Air, const double *Aval,
                         const int *Arp, const int *Aci, const double *Arv, const int *Hjc, const int *Hir,
                         const double *Hval, const double *x, const double *y, double *Ax, double *ATy,
                         double *Hx, RsqpKktArgs a) {
    __shared__ double sh[4];
    small_products_body(desc, Ajc, Air, Aval, Arp, Aci, Arv, Hjc, Hir, Hval, x, y, Ax, ATy, Hx);
    __threadfence_block();
    __syncthreads();
    kkt_body(a, sh);
}

}  // namespace

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
hipError_t rsqp_launch_spmv(const int4 *blkinfo, int nblk, const int *ptr, const int *idx, const double *val,
                            const double *in, double *out, int nbatch, long long ptr_stride,
                            long long nnz_stride, long long in_stride, long long out_stride,
                            hipStream_t stream) {
    if (nblk <= 0 || nbatch <= 0) return hipSuccess;
    const int xcd_map = nbatch >= 8;
    const long long grid = xcd_map ? (long long)((nbatch + 7) / 8) * 8 * nblk : (long long)nbatch * nblk;
    hipLaunchKernelGGL(csx_stream_spmv<false>, dim3((unsigned)grid), dim3(SPMV_NT), 0, stream, blkinfo, nblk, nbatch, xcd_map,
                       ptr, idx, val, in, out, ptr_stride, nnz_stride, in_stride, out_stride, RsqpSpmvDualDx());
    return hipGetLastError();
}
hipError_t rsqp_launch_spmv_dualdx(const int4 *blkinfo, int nblk, const int *ptr, const int *idx, const double *val, const double *in,
                                   double *out, const RsqpSpmvDualDx &epi, hipStream_t stream) {
    if (nblk <= 0) return hipSuccess;
    hipLaunchKernelGGL(csx_stream_spmv<true>, dim3((unsigned)nblk), dim3(SPMV_NT), 0, stream, blkinfo, nblk, 1, 0, ptr, idx, val, in, out, 0LL, 0LL,
                       0LL, 0LL, epi);
    return hipGetLastError();
}

// capacity handed to the block builder: a block may start on an odd entry and is then read
// from the even entry before it, so one slot of the staging array stays in reserve
hipError_t rsqp_launch_spmv_ldsvec(int variant, int nminor, int nslices, const int *slice, const int *ptr,
                                   const int *idx, const unsigned short *idx16, const double *val, const double *in, double *out, int nbatch,
                                   long long ptr_stride, long long nnz_stride, long long in_stride,
                                   long long out_stride, hipStream_t stream) {
    const size_t lds = (size_t)nminor * 8 + 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define LV_LAUNCH(G, U)                                                                                        \
    do {                                                                                                       \
        static std::atomic<unsigned long long> set_{0};                                                        \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_spmv<G, U>), set_, 160 * 1024);         \
        hipLaunchKernelGGL((csx_ldsvec_spmv<G, U>), dim3(nbatch * nslices), dim3(LV_NT), lds, stream, nminor,  \
                           nslices, slice, ptr, idx, val, in, out, ptr_stride, nnz_stride, in_stride,          \
                           out_stride);                                                                        \
    } while (0)
    switch (variant) {
    case 1: LV_LAUNCH(8, 2); break;
    case 2: LV_LAUNCH(8, 4); break;
    case 3: LV_LAUNCH(4, 2); break;
    case 4: LV_LAUNCH(4, 4); break;
    case 5: LV_LAUNCH(16, 2); break;
    case 6: LV_LAUNCH(16, 1); break;
    case 7: LV_LAUNCH(8, 3); break;
    case 8: LV_LAUNCH(2, 4); break;
#define LP_LAUNCH(G, U, N)                                                                                        \
    do {                                                                                                       \
        static std::atomic<unsigned long long> set_{0};                                                        \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_spmv_pipe<G, U, N>), set_, 160 * 1024); \
        hipLaunchKernelGGL((csx_ldsvec_spmv_pipe<G, U, N>), dim3(nbatch * nslices), dim3(LV_NT), lds, stream,     \
                           nminor, nslices, slice, ptr, idx, val, in, out, ptr_stride, nnz_stride, in_stride,  \
                           out_stride);                                                                        \
    } while (0)
    case 11: LP_LAUNCH(8, 2, false); break;
    case 12: LP_LAUNCH(8, 4, false); break;
    case 13: LP_LAUNCH(4, 2, false); break;
    case 14: LP_LAUNCH(4, 4, false); break;
    case 15: LP_LAUNCH(8, 3, false); break;
    case 16: LP_LAUNCH(4, 3, false); break;
    case 17: LP_LAUNCH(8, 6, false); break;
    case 18: LP_LAUNCH(8, 8, false); break;
    case 19: LP_LAUNCH(4, 6, false); break;
    case 22: LP_LAUNCH(8, 4, true); break;
    case 24: LP_LAUNCH(4, 4, true); break;
    case 27: LP_LAUNCH(8, 6, true); break;
    case 29: LP_LAUNCH(4, 6, true); break;
#undef LP_LAUNCH
#define L2_LAUNCH(G, U)                                                                                        \
    do {                                                                                                       \
        if (idx16) {                                                                                           \
            static std::atomic<unsigned long long> set_{0};                                                    \
            rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_spmv_pipe2<G, U, unsigned short>), set_, 160 * 1024); \
            hipLaunchKernelGGL((csx_ldsvec_spmv_pipe2<G, U, unsigned short>), dim3(nbatch * nslices),          \
                               dim3(LV_NT), lds, stream, nminor, nslices, slice, ptr, idx16, val, in, out,     \
                               ptr_stride, nnz_stride, in_stride, out_stride);                                 \
        } else {                                                                                               \
            static std::atomic<unsigned long long> set_{0};                                                    \
            rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_spmv_pipe2<G, U, int>), set_, 160 * 1024); \
            hipLaunchKernelGGL((csx_ldsvec_spmv_pipe2<G, U, int>), dim3(nbatch * nslices), dim3(LV_NT), lds,   \
                               stream, nminor, nslices, slice, ptr, idx, val, in, out, ptr_stride, nnz_stride, \
                               in_stride, out_stride);                                                         \
        }                                                                                                      \
    } while (0)
    case 31: L2_LAUNCH(2, 2); break;
    case 32: L2_LAUNCH(2, 3); break;
    case 33: L2_LAUNCH(4, 1); break;
    case 34: L2_LAUNCH(4, 2); break;
    case 35: L2_LAUNCH(4, 3); break;
    case 36: L2_LAUNCH(8, 1); break;
    case 37: L2_LAUNCH(8, 2); break;
    case 38: L2_LAUNCH(2, 4); break;
#undef L2_LAUNCH
#ifdef RSQP_SPMV_EXPERIMENT
#define LG_LAUNCH(V, W)                                                                                                \
    case V: {                                                                                                          \
        static std::atomic<unsigned long long> s5{0};                                                                  \
        if (!idx16) return hipErrorInvalidValue;                                                                       \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_lanegroup_bound<W>), s5, 160 * 1024);           \
        hipLaunchKernelGGL(csx_ldsvec_lanegroup_bound<W>, dim3(nbatch), dim3(LV_NT), lds, stream, nminor,              \
                           (int)nnz_stride, idx16, val, in, out, nnz_stride, in_stride, out_stride);                   \
        break;                                                                                                         \
    }
    LG_LAUNCH(52, 0) LG_LAUNCH(53, 1) LG_LAUNCH(54, 2) LG_LAUNCH(55, 3) LG_LAUNCH(56, 4)
#undef LG_LAUNCH
#define LG2_LAUNCH(V, W)                                                                                               \
    case V: {                                                                                                          \
        static std::atomic<unsigned long long> s6{0};                                                                  \
        if (!idx16) return hipErrorInvalidValue;                                                                       \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_lanegroup_bound2<W>), s6, 160 * 1024);          \
        hipLaunchKernelGGL(csx_ldsvec_lanegroup_bound2<W>, dim3(nbatch), dim3(LV_NT), lds, stream, nminor,             \
                           (int)nnz_stride, idx16, val, in, out, nnz_stride, in_stride, out_stride);                   \
        break;                                                                                                         \
    }
    LG2_LAUNCH(62, 0) LG2_LAUNCH(66, 4)
#undef LG2_LAUNCH
    case 50: case 51: {
        static std::atomic<unsigned long long> s4{0}, s8{0};
        if (!idx16) return hipErrorInvalidValue;
        if (variant == 50) {
            rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_streambound<4>), s4, 160 * 1024);
            hipLaunchKernelGGL((csx_ldsvec_streambound<4>), dim3(nbatch), dim3(LV_NT), lds, stream, nminor, (int)nnz_stride, idx16, val, in, out, nnz_stride, in_stride, out_stride);
        } else {
            rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_streambound<8>), s8, 160 * 1024);
            hipLaunchKernelGGL((csx_ldsvec_streambound<8>), dim3(nbatch), dim3(LV_NT), lds, stream, nminor, (int)nnz_stride, idx16, val, in, out, nnz_stride, in_stride, out_stride);
        }
        break;
    }
#endif
    default: return hipErrorInvalidValue;
    }
#undef LV_LAUNCH
    return hipGetLastError();
}

hipError_t rsqp_launch_spmv_segscan(int nminor, const int4 *chunks, int nchunks, const int *nzlist, const int *empties,
                                    int nempty, const unsigned *sbits, const unsigned short *idx16, const double *val,
                                    const double *in, double *out, int nbatch, long long nnz_stride, long long in_stride,
                                    long long out_stride, hipStream_t stream) {
    size_t lds = (size_t)((nminor + 1) & ~1) * 8;
    if (lds > 160 * 1024 || !idx16 || nchunks <= 0) return hipErrorInvalidValue;
    const size_t stage = (size_t)(LV_NT / 64) * 128 * 8;          // 1 KB per wave, when the vector leaves room
    const bool st = lds + stage <= 160 * 1024;
    if (st) lds += stage;
    static std::atomic<unsigned long long> set_[4] = {{0}, {0}, {0}, {0}};
    SegPlanView sp;
    sp.chunks = chunks; sp.nzlist = nzlist; sp.empties = empties; sp.nchunks = nchunks; sp.nempty = nempty;
#define SEG_LAUNCH(D, S)                                                                                              \
    do {                                                                                                              \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&csx_ldsvec_segscan<D, S>), set_[2 * D + S], 160 * 1024);  \
        hipLaunchKernelGGL((csx_ldsvec_segscan<D, S>), dim3(nbatch), dim3(LV_NT), lds, stream, nminor, sp,            \
                           reinterpret_cast<const unsigned char *>(sbits), idx16, val, in, out, nnz_stride, in_stride, \
                           out_stride);                                                                               \
    } while (0)
    if (nempty == 0) { if (st) SEG_LAUNCH(true, true); else SEG_LAUNCH(true, false); }
    else             { if (st) SEG_LAUNCH(false, true); else SEG_LAUNCH(false, false); }
#undef SEG_LAUNCH
    return hipGetLastError();
}

int rsqp_spmv_chunk(void) { return SPMV_CHUNK - 2; }

hipError_t rsqp_launch_scatter(int n, const int *order, const int *tmap, const double *tv, double *val,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_values, dim3((n + 255) / 256), dim3(256), 0, stream, n, order, tmap, tv, val);
    return hipGetLastError();
}

hipError_t rsqp_launch_scatter_csc_csr(int n, const int *order, const int *rorder, const double *tv, double *val, double *rval,
                                       hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_values_csc_csr, dim3((n + 255) / 256), dim3(256), 0, stream, n, order, rorder, tv, val, rval);
    return hipGetLastError();
}

namespace {
// the CSR values of a FULLY dense matrix are the transpose of its CSC values: 32 x 32 tiles through LDS instead of a gather through
// the permutation (whose reads are nrow * 8 bytes apart: 28 ms for the 8.4 M entries of the dense 2048 x 4096 configuration)
__global__ void dense_values_transpose(int nrow, int ncol, const double *__restrict__ src, double *__restrict__ dst) {
    __shared__ double tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int r = r0 + threadIdx.x, c = c0 + k;
        tile[k][threadIdx.x] = (r < nrow && c < ncol) ? src[(long long)c * nrow + r] : 0.0;      // tile[c][r]
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int c = c0 + threadIdx.x, r = r0 + k;
        if (r < nrow && c < ncol) dst[(long long)r * ncol + c] = tile[threadIdx.x][k];
    }
}
}  // namespace
hipError_t rsqp_launch_gather_dense(int nrow, int ncol, const double *src, double *dst, hipStream_t stream) {
    if (nrow <= 0 || ncol <= 0) return hipSuccess;
    hipLaunchKernelGGL(dense_values_transpose, dim3((nrow + 31) / 32, (ncol + 31) / 32), dim3(32, 8), 0, stream, nrow, ncol, src, dst);
    return hipGetLastError();
}
hipError_t rsqp_launch_gather(int n, const int *perm, const double *src, double *dst, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_values, dim3((n + 255) / 256), dim3(256), 0, stream, n, perm, src, dst);
    return hipGetLastError();
}

hipError_t rsqp_launch_densify(int nrow, int ncol, const int *jc, const int *ir, const double *val, double *dense,
                               hipStream_t stream) {
    hipError_t e = hipMemsetAsync(dense, 0, sizeof(double) * (size_t)nrow * ncol, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(densify_csc, dim3(ncol), dim3(256), 0, stream, ncol, (long long)nrow, jc, ir, val, dense);
    return hipGetLastError();
}

hipError_t rsqp_launch_kkt(const RsqpKktArgs &a, int nq, hipStream_t stream) {
    hipLaunchKernelGGL(kkt_kernel, dim3(nq), dim3(KKT_NT), 0, stream, a);
    return hipGetLastError();
}

hipError_t rsqp_launch_small_certificate(const QPPools &p, const RsqpKktArgs &a, int nq, double *Ax, double *ATy,
                                         double *Hx, hipStream_t stream) {
    hipLaunchKernelGGL(small_certificate_kernel, dim3(nq), dim3(KKT_NT), 0, stream, p.desc, p.Ajc, p.Air, p.Aval,
                       p.Arp, p.Aci, p.Arv, p.Hjc, p.Hir, p.Hval, p.x, p.y, Ax, ATy, Hx, a);
    return hipGetLastError();
}


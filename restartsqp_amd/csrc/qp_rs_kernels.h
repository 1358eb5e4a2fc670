// qp_rs_kernels.h -- kernels of the GENERAL range-space path of the HBM-resident engine (qp_large.hip, Impl::rsh):
// any symmetric positive definite Hessian. Included inside qp_large.hip's anonymous namespace (it uses lane_sum4, block_sum,
// publish, delta_of, NT, SYT of that file).
//
// Formulation. The active BOUNDS and the active CONSTRAINTS are both rows of one matrix C (row id v < nV: e_v', the bound of
// variable v; id nV + i: row i of A), and the engine keeps the explicit inverse of the Schur complement
//
//     S = C H^-1 C'   (nR x nR, nR = fixed variables + active constraints),        Sinv = S^-1, upper triangle (k_sym_tile)
//
// H never changes during a solve, so H^-1 is an OPERATOR built once per Hessian: the LDL' factor of a banded H (half bandwidth
// <= 2) applied by ONE workgroup in chunked form (k_band_apply), or the explicit dense inverse (GEMV). Every working-set change is
// one of TWO operations -- a row joins (bordering of Sinv) or leaves (rank-1 elimination) -- whatever it is a row of.
//
//     step direction:  S dl = db + C p  (p = H^-1 (gN - g)),   q = C'dl,   dx = H^-1 (q - (gN - g)),   H dx = q - (gN - g)
//     row c joins:     w = H^-1 c',  cv = C w,  u = Sinv cv,  s = c w - cv'u;  Sinv <- [[Sinv + u u'/s, -u/s], [-u'/s, 1/s]]
//     row k leaves:    Sinv -= v v'/v_k (v = column k), the last row / column moves into the slot
//
// Reference: the arithmetic stands in for qpOASES' SQProblem::hotstart / init behind src/qpOASESInterface.cpp:155-206.
#pragma once

// ---- banded H^-1 (half bandwidth <= 2) ----------------------------------------------------------------------------------------------
// H = L D L' (L unit lower, sub-diagonals l1, l2), S = D^1/2, M = S^-1 L S (unit lower as well):  H^-1 = S^-1 M^-T M^-1 S^-1. The
// diagonal scalings ride on the coalesced load / store of the vector; the two triangular solves are linear recurrences of depth 2,
// i.e. compositions of affine maps of the state (z[i-1], z[i-2]). 1024 threads; thread t owns the c = ceil(n / 1024) consecutive
// rows of chunk t with their factor entries in REGISTERS (loaded once, up front, from a chunk-interleaved layout: consecutive
// threads read consecutive addresses):
//   1. the chunk's map "incoming state -> outgoing state" (a particular and two homogeneous runs over the c rows),
//   2. an inclusive scan of the 1024 maps,
//   3. the chunk again with its true incoming state.
// Forward, then the same backward (reversed order): ~4 c + 12 dependent steps per sweep instead of 2 n.
//   k_band_apply<C>     ONE workgroup of 1024 threads (scan: 6 shuffle steps per wave, the 16 wave totals through LDS); a grid of
//                       them for the columns of a matrix (blocked set-up).
//   k_band_apply_mw<C>  ONE vector on 16 single-wave workgroups: a workgroup moves 1/16 of the bytes (a single compute unit
//                       streams ~60-100 GB/s: the one-workgroup kernel took 23 us for the ~0.7 MB of a product at n = 10 000,
//                       profiles/r05_e_kernel_stats_large_band5.csv) and hands its total map to the others through
//                       device-scope words tagged with the call's sequence number (forward: to the workgroups above it,
//                       backward: below). All 16 are resident at once, every spin is bounded.
// History: the first version (one thread chaining 128 chunk states, factor entries fetched inside the dependent loop) took 62 us.
struct BandOp {
    int n, c;                     // rows, rows per chunk (1024 c >= n)
    const double *m1i, *m2i;      // m1i[k * 1024 + t] = M[i][i-1] for i = t c + k, k = 0..c;  m2i: M[i][i-2], k = 0..c + 1  (0 beyond n)
    const double *sinv;           // 1 / sqrt(d), plain
    double *agg;                  // [2][16][8] totals of the multi-workgroup form: 6 doubles of the map + the tag
    int *err;                     // set when a spin ran out (never seen; the engine then reports a set-up failure)
};
constexpr int BAND_MAX_N = 16384;         // the vector passes through LDS (128 KB)
constexpr int BAND_TH = 1024;
constexpr int BAND_MW = 16;               // workgroups of the multi-workgroup form (one wave each)
struct AMap { double p00, p01, p10, p11, e0, e1; };       // s_out = P s_in + e
__device__ __forceinline__ AMap amap_then(const AMap &a, const AMap &b) {      // first a, then b
    AMap r;
    r.p00 = b.p00 * a.p00 + b.p01 * a.p10; r.p01 = b.p00 * a.p01 + b.p01 * a.p11;
    r.p10 = b.p10 * a.p00 + b.p11 * a.p10; r.p11 = b.p10 * a.p01 + b.p11 * a.p11;
    r.e0 = b.p00 * a.e0 + b.p01 * a.e1 + b.e0; r.e1 = b.p10 * a.e0 + b.p11 * a.e1 + b.e1;
    return r;
}
__device__ __forceinline__ AMap amap_shfl(const AMap &m, int src) {
    AMap r;
    r.p00 = __shfl(m.p00, src); r.p01 = __shfl(m.p01, src); r.p10 = __shfl(m.p10, src); r.p11 = __shfl(m.p11, src);
    r.e0 = __shfl(m.e0, src); r.e1 = __shfl(m.e1, src);
    return r;
}
__device__ __forceinline__ void amap_apply(const AMap &q, double &a, double &b) {
    const double na = q.p00 * a + q.p01 * b + q.e0, nb = q.p10 * a + q.p11 * b + q.e1;
    a = na; b = nb;
}
// inclusive scan of the maps of one wave in processing order (rev: from lane 63 down)
__device__ __forceinline__ AMap band_wave_scan(AMap m, bool rev) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int src = rev ? lane + d : lane - d;
        const AMap o = amap_shfl(m, src & 63);
        if (rev ? (lane + d < 64) : (lane >= d)) m = amap_then(o, m);
    }
    return m;
}
// the chunk's map (forward: rows 0..cnt-1 with M[i][i-1], M[i][i-2]; backward: rows cnt-1..0 with M[i+1][i], M[i+2][i])
template <int C, bool REV>
__device__ __forceinline__ AMap band_chunk_map(const double (&b)[C], const double (&L1)[C + 1], const double (&L2)[C + 2], int cnt) {
    double z1 = 0.0, z2 = 0.0, a1 = 1.0, a2 = 0.0, b1 = 0.0, b2 = 1.0;
#pragma unroll
    for (int kk = 0; kk < C; kk++) {
        const int k = REV ? C - 1 - kk : kk;
        if (k < cnt) {
            const double f1 = REV ? L1[k + 1] : L1[k], f2 = REV ? L2[k + 2] : L2[k];
            const double z = b[k] - f1 * z1 - f2 * z2, na = -f1 * a1 - f2 * a2, nb = -f1 * b1 - f2 * b2;
            z2 = z1; z1 = z; a2 = a1; a1 = na; b2 = b1; b1 = nb;
        }
    }
    AMap m;
    m.p00 = a1; m.p10 = a2; m.p01 = b1; m.p11 = b2; m.e0 = z1; m.e1 = z2;
    return m;
}
template <int C, bool REV>
__device__ __forceinline__ void band_chunk_run(double (&b)[C], const double (&L1)[C + 1], const double (&L2)[C + 2], int cnt, double t1, double t2) {
#pragma unroll
    for (int kk = 0; kk < C; kk++) {
        const int k = REV ? C - 1 - kk : kk;
        if (k < cnt) {
            const double f1 = REV ? L1[k + 1] : L1[k], f2 = REV ? L2[k + 2] : L2[k];
            const double z = b[k] - f1 * t1 - f2 * t2;
            t2 = t1; t1 = z; b[k] = z;
        }
    }
}
// the step direction's product in one launch: in = q - (gN - g) with q = A'dl_C + dl_B formed on the way (and stored: it IS H dx),
// out = dx on the FREE variables only (the fixed ones keep the move of their bound, which dx holds on entry)
struct BandQ { const int *Sb; const double *ATdy, *dy, *gN, *g; double *Hdx; };
__device__ __forceinline__ double band_input(const BandQ &q, const double *__restrict__ in, const double *__restrict__ sub, int i) {
    if (q.Sb) { const double h = (q.ATdy[i] + (q.Sb[i] != 0 ? q.dy[i] : 0.0)) - (q.gN[i] - q.g[i]); q.Hdx[i] = h; return h; }
    return sub ? in[i] - sub[i] : in[i];
}
// out = H^-1 (in - sub)   (sub may be null; in == out allowed). One workgroup; column blockIdx.x of a matrix when ld != 0.
template <int C>
__global__ void __launch_bounds__(BAND_TH) k_band_apply(BandOp op, const double *__restrict__ in, const double *__restrict__ sub,
                                                        double *__restrict__ out, long long ld, BandQ q) {
    extern __shared__ double band_lds[];
    __shared__ AMap wm[BAND_TH / 64];
    double *v = band_lds;
    const int n = op.n, c = op.c, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long off = (long long)blockIdx.x * ld;
    in += off; out += off;
    double L1[C + 1], L2[C + 2];
#pragma unroll
    for (int k = 0; k < C + 1; k++) L1[k] = k <= c ? op.m1i[k * BAND_TH + tid] : 0.0;
#pragma unroll
    for (int k = 0; k < C + 2; k++) L2[k] = k <= c + 1 ? op.m2i[k * BAND_TH + tid] : 0.0;
    for (int i = tid; i < n; i += BAND_TH) v[i] = band_input(q, in, sub, i) * op.sinv[i];
    __syncthreads();
    const int s0i = tid * c, cnt = max(0, min(c, n - s0i));
    double b[C];
#pragma unroll
    for (int k = 0; k < C; k++) b[k] = k < cnt ? v[s0i + k] : 0.0;
#pragma unroll
    for (int sweep = 0; sweep < 2; sweep++) {
        const bool rev = sweep == 1;
        AMap m = rev ? band_chunk_map<C, true>(b, L1, L2, cnt) : band_chunk_map<C, false>(b, L1, L2, cnt);
        m = band_wave_scan(m, rev);
        __syncthreads();                                   // (wm may still be read by the previous sweep)
        if (lane == (rev ? 0 : 63)) wm[wave] = m;
        __syncthreads();
        double a = 0.0, bb = 0.0;                          // state entering this wave
        if (!rev) { for (int w = 0; w < wave; w++) amap_apply(wm[w], a, bb); }
        else { for (int w = BAND_TH / 64 - 1; w > wave; w--) amap_apply(wm[w], a, bb); }
        const AMap pv = amap_shfl(m, (rev ? lane + 1 : lane - 1) & 63);     // the map up to the chunk before this one, inside the wave
        if (!(rev ? (lane == 63) : (lane == 0))) amap_apply(pv, a, bb);
        if (rev) band_chunk_run<C, true>(b, L1, L2, cnt, a, bb); else band_chunk_run<C, false>(b, L1, L2, cnt, a, bb);
    }
#pragma unroll
    for (int k = 0; k < C; k++) if (k < cnt) v[s0i + k] = b[k];
    __syncthreads();
    if (q.Sb) { for (int i = tid; i < n; i += BAND_TH) if (q.Sb[i] == 0) out[i] = v[i] * op.sinv[i]; }
    else for (int i = tid; i < n; i += BAND_TH) out[i] = v[i] * op.sinv[i];
}
// the same product of ONE vector on BAND_MW single-wave workgroups (see above); seq: this call's tag (> 0, increasing)
template <int C>
__global__ void __launch_bounds__(64) k_band_apply_mw(BandOp op, const double *__restrict__ in, const double *__restrict__ sub,
                                                      double *__restrict__ out, BandQ q, double seq) {
    __shared__ double v[64 * C];
    __shared__ double wa[BAND_MW][6];
    const int n = op.n, c = op.c, lane = threadIdx.x, blk = blockIdx.x, tid = blk * 64 + lane;
    double L1[C + 1], L2[C + 2];
#pragma unroll
    for (int k = 0; k < C + 1; k++) L1[k] = k <= c ? op.m1i[k * BAND_TH + tid] : 0.0;
#pragma unroll
    for (int k = 0; k < C + 2; k++) L2[k] = k <= c + 1 ? op.m2i[k * BAND_TH + tid] : 0.0;
    const int r0 = blk * 64 * c, r1 = min(r0 + 64 * c, n);          // the rows of this workgroup
    for (int i = r0 + lane; i < r1; i += 64) v[i - r0] = band_input(q, in, sub, i) * op.sinv[i];
    __syncthreads();
    const int s0i = tid * c, cnt = max(0, min(c, n - s0i));
    double b[C];
#pragma unroll
    for (int k = 0; k < C; k++) b[k] = k < cnt ? v[s0i - r0 + k] : 0.0;
#pragma unroll
    for (int sweep = 0; sweep < 2; sweep++) {
        const bool rev = sweep == 1;
        AMap m = rev ? band_chunk_map<C, true>(b, L1, L2, cnt) : band_chunk_map<C, false>(b, L1, L2, cnt);
        m = band_wave_scan(m, rev);
        double *mine = op.agg + ((size_t)sweep * BAND_MW + blk) * 8;
        if (lane == (rev ? 0 : 63)) {                      // this workgroup's total: six words, then the tag behind a device-scope fence
            __hip_atomic_store(mine + 0, m.p00, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(mine + 1, m.p01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 2, m.p10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(mine + 3, m.p11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 4, m.e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(mine + 5, m.e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            __hip_atomic_store(mine + 6, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lane w fetches the total of workgroup w if that one is processed before this one
        const bool need = lane < BAND_MW && (rev ? lane > blk : lane < blk);
        if (need) {
            const double *src = op.agg + ((size_t)sweep * BAND_MW + lane) * 8;
            int spins = 0;
            while (__hip_atomic_load(src + 6, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { atomicExch(op.err, 1); break; }
            }
            __threadfence();
#pragma unroll
            for (int e = 0; e < 6; e++) wa[lane][e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        double a = 0.0, bb = 0.0;                          // state entering this workgroup
        if (!rev) { for (int w = 0; w < blk; w++) { AMap t; t.p00 = wa[w][0]; t.p01 = wa[w][1]; t.p10 = wa[w][2]; t.p11 = wa[w][3]; t.e0 = wa[w][4]; t.e1 = wa[w][5]; amap_apply(t, a, bb); } }
        else { for (int w = BAND_MW - 1; w > blk; w--) { AMap t; t.p00 = wa[w][0]; t.p01 = wa[w][1]; t.p10 = wa[w][2]; t.p11 = wa[w][3]; t.e0 = wa[w][4]; t.e1 = wa[w][5]; amap_apply(t, a, bb); } }
        const AMap pv = amap_shfl(m, (rev ? lane + 1 : lane - 1) & 63);
        if (!(rev ? (lane == 63) : (lane == 0))) amap_apply(pv, a, bb);
        if (rev) band_chunk_run<C, true>(b, L1, L2, cnt, a, bb); else band_chunk_run<C, false>(b, L1, L2, cnt, a, bb);
        __syncthreads();                                   // (wa is rewritten by the next sweep)
    }
#pragma unroll
    for (int k = 0; k < C; k++) if (k < cnt) v[s0i - r0 + k] = b[k];
    __syncthreads();
    for (int i = r0 + lane; i < r1; i += 64) if (!q.Sb || q.Sb[i] == 0) out[i] = v[i - r0] * op.sinv[i];
}

// (Tried and dropped in round 5, profiles/r05_l_kernel_stats_sparse.csv / r05_m_*: cv = C w over the active rows by one thread per row
//  instead of a full product with A + a gather -- four dependent memory hops per thread, 13 us against 5.9 + 3.9 us; stage 1 of the
//  independence test and the carried step's dot product in the tail of the reduction kernel (last-ticket workgroup) -- 24.6 us
//  against 9.0 + 7.7 us for the two launches: a dependent kernel boundary costs ~1.5 us on this chip, a one-workgroup tail behind a
//  grid-wide ticket more. Cold start of the sparse configuration 1.99 s with both against 2.01 s without: reverted.)
// symmetry (pattern and values) and half bandwidth of H from its CSC, on the device: flag[0] = max |row - column|, flag[1] != 0 when an
// entry has no mirror image (1) or a different one (2). One workgroup per column; the mirror entry by binary search in the sorted
// column. (The host loop this replaces took 0.3 s on the 4.2 M entries of a dense 2048 x 2048 Hessian -- a quarter of that
// configuration's cold start.)
__global__ void __launch_bounds__(NT) k_rs_sym_check(int nV, const int *__restrict__ jc, const int *__restrict__ ir, const double *__restrict__ val,
                                                     int *__restrict__ flag) {
    const int c = blockIdx.x;
    int hb = 0, bad = 0;
    for (int k = jc[c] + threadIdx.x; k < jc[c + 1]; k += NT) {
        const int r = ir[k];
        hb = max(hb, abs(r - c));
        if (r == c) continue;
        int lo = jc[r], hi = jc[r + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ir[mid] < c) lo = mid + 1; else hi = mid; }
        if (lo >= jc[r + 1] || ir[lo] != c) bad |= 1;
        else { const double a = val[k], b = val[lo]; if (fabs(a - b) > 1e-12 * fmax(fabs(a), fabs(b))) bad |= 2; }
    }
    for (int o = 32; o > 0; o >>= 1) { hb = max(hb, __shfl_xor(hb, o)); bad |= __shfl_xor(bad, o); }
    if ((threadIdx.x & 63) == 0) { if (hb > 0) atomicMax(flag, hb); if (bad) atomicOr(flag + 1, bad); }
}

// ---- LAZY rank-1 updates of Sinv (round 5) ----------------------------------------------------------------------------------------
// The effective matrix is  Sinv = M + sum_k c_k p_k p_k'  with up to LZK pending rank-1 terms (the bordering of a row that joined:
// p = u, c = 1 / s; the elimination of a row that left: p = its column, c = -1 / v_j). A product never writes M -- it reads the
// triangle once (k_sym_tile<false, true>) and adds sum_k c_k (p_k'w) p_k in the reduction -- and one pass applies all pending
// terms when the list is full (k_sym_tile_lz). Before: every row that joined or left cost a read + WRITE pass over the triangle
// (0.2 GB at nR = 5 000); now a row that leaves costs O(nR), one that joins a read-only pass, and the write pass comes every
// LZK changes.
constexpr int LZK = 6;
struct LzP { const double *vec; long long stride; const double *c; int np; };
__global__ void __launch_bounds__(NT) k_lz_dots(LzP P, int n, const double *__restrict__ w, double *__restrict__ d) {
    __shared__ double sh[4];
    const double *p = P.vec + (long long)blockIdx.x * P.stride;
    double s = lane_sum4(n, [&](int i) { return p[i] * w[i]; });
    s = block_sum(s, sh);
    if (threadIdx.x == 0) d[blockIdx.x] = s;
}
__global__ void k_sym_reduce_lz(int n, int nt, const double *__restrict__ P1, const double *__restrict__ P2, double *__restrict__ y,
                                const int *__restrict__ R, double *__restrict__ full, LzP P, const double *__restrict__ d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int Ti = i / SYT;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int J = Ti;
    for (; J + 3 < nt; J += 4) {
        a0 += P1[(long long)J * n + i]; a1 += P1[(long long)(J + 1) * n + i]; a2 += P1[(long long)(J + 2) * n + i]; a3 += P1[(long long)(J + 3) * n + i];
    }
    for (; J < nt; J++) a0 += P1[(long long)J * n + i];
    int I = 0;
    for (; I + 3 <= Ti; I += 4) {
        b0 += P2[(long long)I * n + i]; b1 += P2[(long long)(I + 1) * n + i]; b2 += P2[(long long)(I + 2) * n + i]; b3 += P2[(long long)(I + 3) * n + i];
    }
    for (; I <= Ti; I++) b0 += P2[(long long)I * n + i];
    double val = ((a0 + a1) + (a2 + a3)) + ((b0 + b1) + (b2 + b3));
    for (int k = 0; k < P.np; k++) val += (P.c[k] * d[k]) * P.vec[(long long)k * P.stride + i];
    y[i] = val;
    if (R) full[R[i]] = val;
}
// M += sum_k c_k p_k p_k' on the upper tiles (the flush)
__global__ void __launch_bounds__(256) k_sym_tile_lz(double *__restrict__ M, long long ld, int n, LzP P) {
    __shared__ double pj[LZK][SYT];       // the terms' entries of this tile's columns (every lane needs all 64 of them)
    int I, J;
    sym_tile_of((int)blockIdx.x, I, J);
    const bool diag = I == J;
    const int ii = threadIdx.x & 63, wv = threadIdx.x >> 6, i = I * SYT + ii;
    for (int e = threadIdx.x; e < LZK * SYT; e += 256) {
        const int k = e / SYT, jj = e % SYT, j = J * SYT + jj;
        pj[k][jj] = (k < P.np && j < n) ? P.vec[(long long)k * P.stride + j] : 0.0;
    }
    if ((ld & 1) == 0 && (reinterpret_cast<unsigned long long>(M) & 15) == 0) {
        // 16 bytes per lane (two rows of a column, as k_sym_tile's read-only product): lane pair-of-rows r2 of column 8 c + (t >> 5).
        // An entry outside the triangle (below the diagonal of a diagonal tile, behind row n) is written back as it was read
        const int r2 = threadIdx.x & 31, cg = threadIdx.x >> 5, i0 = I * SYT + 2 * r2;
        double p0[LZK], p1[LZK];
#pragma unroll
        for (int k = 0; k < LZK; k++) {
            p0[k] = (k < P.np && i0 < n) ? P.c[k] * P.vec[(long long)k * P.stride + i0] : 0.0;
            p1[k] = (k < P.np && i0 + 1 < n) ? P.c[k] * P.vec[(long long)k * P.stride + i0 + 1] : 0.0;
        }
        __syncthreads();
        double2 m2[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int j = J * SYT + c * 8 + cg;
            m2[c] = (i0 < n && j < n) ? *reinterpret_cast<const double2 *>(M + (long long)j * ld + i0) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int jj = c * 8 + cg, j = J * SYT + jj;
            if (i0 < n && j < n && (!diag || 2 * r2 <= jj)) {
                double2 m = m2[c];
                const bool v1 = i0 + 1 < n && (!diag || 2 * r2 + 1 <= jj);
#pragma unroll
                for (int k = 0; k < LZK; k++) { m.x += p0[k] * pj[k][jj]; m.y = v1 ? m.y + p1[k] * pj[k][jj] : m.y; }
                *reinterpret_cast<double2 *>(M + (long long)j * ld + i0) = m;
            }
        }
        return;
    }
    double pi[LZK];
#pragma unroll
    for (int k = 0; k < LZK; k++) pi[k] = (k < P.np && i < n) ? P.c[k] * P.vec[(long long)k * P.stride + i] : 0.0;
    __syncthreads();
    double mm[16];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const int jj = wv * 16 + c, j = J * SYT + jj;
        mm[c] = (i < n && j < n && (!diag || ii <= jj)) ? M[(long long)j * ld + i] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const int jj = wv * 16 + c, j = J * SYT + jj;
        if (i < n && j < n && (!diag || ii <= jj)) {
            double m = mm[c];
#pragma unroll
            for (int k = 0; k < LZK; k++) m += pi[k] * pj[k][jj];
            M[(long long)j * ld + i] = m;
        }
    }
}
// column j of the EFFECTIVE matrix and the coefficient of its elimination (scal[9] = -1 / v_j)
__global__ void k_lz_colcoef(const double *__restrict__ M, long long ld, int n, int j, LzP P, double *__restrict__ v, double *__restrict__ scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double val = i <= j ? M[(long long)j * ld + i] : M[(long long)i * ld + j];
    for (int k = 0; k < P.np; k++) { const double *p = P.vec + (long long)k * P.stride; val += P.c[k] * p[j] * p[i]; }
    v[i] = val;
    if (i == j) scal[9] = val != 0.0 ? -1.0 / val : 0.0;
}
// a row joined at position n (after the products: u = Sinv cv, 1 / s in scal[8]): column n of M = -u / s, corner 1 / s, working-set
// entry, multiplier; the rank-1 part u u' / s becomes pending term `slot` (zero behind n), the older terms get a zero at n
__global__ void k_lz_border(double *M, long long ld, int n, const double *__restrict__ u, const double *__restrict__ scal, int *R, int *posR,
                            int *Sall, int id, int side, double *y, int yidx, double yval, double *pvec, long long stride, int slot,
                            double *pc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    const double is = scal[8];
    double *pn = pvec + (long long)slot * stride;
    if (j == n) {
        R[n] = id; posR[id] = n; Sall[id] = side;
        if (yidx >= 0) y[yidx] = yval;
        M[(long long)n * ld + n] = is;
        for (int k = 0; k <= slot; k++) { pvec[(long long)k * stride + n] = 0.0; pvec[(long long)k * stride + n + 1] = 0.0; }
        pc[slot] = is;
    } else {
        const double uj = u[j];
        M[(long long)n * ld + j] = -uj * is;
        pn[j] = uj;
    }
}
// the row at position j leaves (v = its effective column, scal[9] = -1 / v_j): new pending term `slot` = v with the LAST entry moved
// into slot j (as the matrix does), the older terms permuted the same way. (k_dual_move_last_sym moves the stored row / column.)
__global__ void k_lz_remove_fix(int n, int j, const double *__restrict__ v, const double *__restrict__ scal, double *pvec, long long stride,
                                int slot, double *pc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int last = n - 1;
    double *pn = pvec + (long long)slot * stride;
    if (i < last) pn[i] = i == j ? v[last] : v[i];
    if (i == 0) {
        pc[slot] = scal[9];
        if (j != last) for (int k = 0; k < slot; k++) pvec[(long long)k * stride + j] = pvec[(long long)k * stride + last];
    }
}

// a rank-1 term c v v' on the current rows joins the pending list (DESIGN 4.4's bound changes: Sherman-Morrison terms)
__global__ void k_lz_push(int n, const double *__restrict__ v, const double *__restrict__ scal, int src, double *pvec, long long stride, int slot,
                          double *pc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pvec[(long long)slot * stride + i] = v[i];
    if (i == 0) pc[slot] = scal[src];
}

__global__ void k_rs_unit_diag(int n, double *__restrict__ X, long long ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[i + (long long)i * ld] = 1.0;
}

// ---- rows of C ----------------------------------------------------------------------------------------------------------------
// the incoming row as a dense vector: id < nV: e_id, else row id - nV of A (all variables). One workgroup.
__global__ void __launch_bounds__(NT) k_rs_row(int nV, int id, const int *__restrict__ rp, const int *__restrict__ ci,
                                               const double *__restrict__ rv, const double *__restrict__ denseAT, double *__restrict__ a) {
    if (id >= nV && denseAT) {
        const double *row = denseAT + (long long)(id - nV) * nV;
        for (int v = threadIdx.x; v < nV; v += NT) a[v] = row[v];
        return;
    }
    for (int v = threadIdx.x; v < nV; v += NT) a[v] = 0.0;
    __syncthreads();
    if (id < nV) { if (threadIdx.x == 0) a[id] = 1.0; return; }
    const int r = id - nV;
    for (int k = rp[r] + threadIdx.x; k < rp[r + 1]; k += NT) a[ci[k]] = rv[k];
}
// w = Hinv c' for the explicit dense inverse: a bound picks column v, a sparse row combines the columns of its entries
__global__ void __launch_bounds__(NT) k_rs_hinv_row(int nV, int id, const int *__restrict__ rp, const int *__restrict__ ci,
                                                    const double *__restrict__ rv, const double *__restrict__ Hinv, long long ld,
                                                    double *__restrict__ w) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= nV) return;
    if (id < nV) { w[i] = Hinv[(long long)id * ld + i]; return; }
    const int r = id - nV;
    double s = 0.0;
    for (int k = rp[r]; k < rp[r + 1]; k++) s += rv[k] * Hinv[(long long)ci[k] * ld + i];
    w[i] = s;
}
// cv[j] = (C w)[j]: w itself on a bound row, (A w) on a constraint row
__global__ void k_rs_gather(int nR, const int *__restrict__ R, int nV, const double *__restrict__ w, const double *__restrict__ Aw,
                            double *__restrict__ cv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nR) return;
    const int id = R[j];
    cv[j] = id < nV ? w[id] : Aw[id - nV];
}
// the reverse: a vector over the active rows back to [variables; constraints] (entries of inactive rows are left alone)
__global__ void k_rs_scatter(int nR, const int *__restrict__ R, const double *__restrict__ act, double *__restrict__ full) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nR) full[R[j]] = act[j];
}
// ... and only the constraint rows, by constraint index (the exchange rule's xi_C)
__global__ void k_rs_scatter_c(int nR, const int *__restrict__ R, int nV, const double *__restrict__ act, double *__restrict__ byc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nR && R[j] >= nV) byc[R[j] - nV] = act[j];
}
// independence of an incoming row c from the active ones. Stage 1 (atu == null): a2 = |c_FR|^2, ad = c w (w = H^-1 c'),
// s = ad - cv'u the pivot of the bordering. Stage 2 (atu = A_AC' u_C): r2 = |(c - atu)_FR|^2 as well -- the first-order residual
// of the row's representation by the active rows (Z'r = Z'c: the test of the other engines). One workgroup of 1024; published.
__global__ void __launch_bounds__(1024) k_rs_li_publish(int nV, const int *__restrict__ Sb, const double *__restrict__ a,
                                                        const double *__restrict__ w, const double *__restrict__ atu, int k,
                                                        const double *__restrict__ cv, const double *__restrict__ u,
                                                        double *__restrict__ scal, double *__restrict__ ctl, double seqv) {
    __shared__ double sh[4][16];
    double a2 = 0.0, r2 = 0.0, ad = 0.0, cu = 0.0;
    for (int i = threadIdx.x; i < nV; i += 1024) {
        const double x0 = a[i];
        ad += x0 * w[i];
        if (Sb[i] == 0) { a2 += x0 * x0; if (atu) { const double q0 = x0 - atu[i]; r2 += q0 * q0; } }
    }
    for (int j = threadIdx.x; j < k; j += 1024) cu += cv[j] * u[j];
    double v4[4] = {a2, r2, ad, cu};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double v = v4[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { double v = 0.0; for (int x = 0; x < 16; x++) v += sh[q][x]; t[q] = v; }
    const double sp = t[2] - t[3];
    scal[6] = t[0]; scal[7] = t[1]; scal[5] = sp; scal[8] = sp != 0.0 ? 1.0 / sp : 0.0;
    ctl[2] = t[0]; ctl[3] = t[1]; ctl[4] = sp; ctl[5] = t[2];
    publish(ctl, seqv);
}

// ---- step direction -------------------------------------------------------------------------------------------------------------
// right-hand side of S dl = db + C p over the active rows
__global__ void k_rs_rhs(int nR, const int *__restrict__ R, int nV, const int *__restrict__ Sall, const double *__restrict__ lb,
                         const double *__restrict__ ub, const double *__restrict__ lbN, const double *__restrict__ ubN,
                         const double *__restrict__ lbA, const double *__restrict__ ubA, const double *__restrict__ lbAN,
                         const double *__restrict__ ubAN, const double *__restrict__ p, const double *__restrict__ Ap,
                         double *__restrict__ rhs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nR) return;
    const int id = R[j];
    if (id < nV) rhs[j] = (Sall[id] == -1 ? delta_of(lbN[id], lb[id]) : delta_of(ubN[id], ub[id])) + p[id];
    else { const int r = id - nV; rhs[j] = (Sall[id] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) + Ap[r]; }
}
// q = A'dl_C + dl_B;  H dx = q - (gN - g)   (dx = H^-1 of it follows)
// (dense operator: the banded one forms this inside k_band_apply)
__global__ void k_rs_q(int nV, const int *__restrict__ Sb, const double *__restrict__ ATdy, const double *__restrict__ dy,
                       const double *__restrict__ gN, const double *__restrict__ g, double *__restrict__ Hdx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
    Hdx[i] = (ATdy[i] + (Sb[i] != 0 ? dy[i] : 0.0)) - (gN[i] - g[i]);
}
// the multiplier step CARRIED over a row that joined at position k (the right-hand sides of the rows that were active before
// have (1 - tau) of their way left): Sinv_new rhs_new = [om dl - lam u; lam], lam = (rhs_k - om cv'dl) / s -- O(nR) instead of a
// pass over Sinv. One workgroup; dy (by row id, zeroed before) is filled as well.
__global__ void __launch_bounds__(1024) k_rs_carry_add(int k, double om, const double *__restrict__ cv, const double *__restrict__ u,
                                                     double *__restrict__ dl, const double *__restrict__ scal, int id, int nV,
                                                     const int *__restrict__ Sall, const double *__restrict__ lb,
                                                     const double *__restrict__ ub, const double *__restrict__ lbN,
                                                     const double *__restrict__ ubN, const double *__restrict__ lbA,
                                                     const double *__restrict__ ubA, const double *__restrict__ lbAN,
                                                     const double *__restrict__ ubAN, const double *__restrict__ p,
                                                     const double *__restrict__ Ap, const int *__restrict__ R, double *__restrict__ dy) {
    __shared__ double sh[16];
    double t = 0.0;
    for (int j = threadIdx.x; j < k; j += 1024) t += cv[j] * dl[j];
    t = block_sum_w(t, sh);
    double rk;
    if (id < nV) rk = (Sall[id] == -1 ? delta_of(lbN[id], lb[id]) : delta_of(ubN[id], ub[id])) + p[id];
    else { const int r = id - nV; rk = (Sall[id] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) + Ap[r]; }
    const double lam = (rk - om * t) * scal[8];
    for (int j = threadIdx.x; j < k; j += 1024) { const double d = om * dl[j] - lam * u[j]; dl[j] = d; dy[R[j]] = d; }
    if (threadIdx.x == 0) { dl[k] = lam; dy[id] = lam; }
}
__global__ void k_rs_diff(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

// ---- small dense problems: the static tableau WW = [I; A] H^-1 [I A']  ((nV + nC)^2, built once per Hessian; rs_kind 3) ----------
// Column id of WW is [w; A w] for the row with that id (w = H^-1 c'): the products of an incoming row, the step direction and the
// set-up matrix are GATHERS and one gathered-column GEMV instead of products with A, A' and H^-1 (dense 2048 x 4096: three
// 32-64 MB products per change less).
// cv[j] = (C w)[j] = WW[R[j]][id]; scal[so] = c w = WW[id][id]
__global__ void k_ww_cv(int nR, const int *__restrict__ R, const double *__restrict__ WW, long long ldw, int id, double *__restrict__ cv,
                        double *__restrict__ scal, int so) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const double *col = WW + (long long)id * ldw;
    if (j < nR) cv[j] = col[R[j]];
    if (j == 0) scal[so] = col[id];
}
// stage 1 of the independence test from the tableau: |c_FR|^2 from the row itself (dense copy or CSR), c w from scal[so],
// cv'u over the active rows. One workgroup; published.
__global__ void __launch_bounds__(NT) k_ww_li_publish(int nV, const int *__restrict__ Sb, int id, const double *__restrict__ denseAT,
                                                      const int *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ rv,
                                                      int k, const double *__restrict__ cv, const double *__restrict__ u,
                                                      double *__restrict__ scal, int so, double *__restrict__ ctl, double seqv) {
    __shared__ double sh[4];
    double a2;
    if (id < nV) a2 = (threadIdx.x == 0 && Sb[id] == 0) ? 1.0 : 0.0;
    else if (denseAT) {
        const double *row = denseAT + (long long)(id - nV) * nV;
        a2 = lane_sum4(nV, [&](int v) { const double x0 = row[v]; return Sb[v] == 0 ? x0 * x0 : 0.0; });
    } else {
        const int r = id - nV, k0 = rp[r], n = rp[r + 1] - k0;
        a2 = lane_sum4(n, [&](int e) { const double x0 = rv[k0 + e]; return Sb[ci[k0 + e]] == 0 ? x0 * x0 : 0.0; });
    }
    double cu = lane_sum4(k, [&](int j) { return cv[j] * u[j]; });
    a2 = block_sum(a2, sh);
    cu = block_sum(cu, sh);
    if (threadIdx.x != 0) return;
    const double ad = scal[so], sp = ad - cu;
    scal[6] = a2; scal[7] = 0.0; scal[5] = sp; scal[8] = sp != 0.0 ? 1.0 / sp : 0.0;
    ctl[2] = a2; ctl[3] = 0.0; ctl[4] = sp; ctl[5] = ad;
    publish(ctl, seqv);
}
// the step direction in ONE launch: [dx; A dx] = sum_j dl[j] WW[:, R[j]] - [p; A p]. A workgroup owns 16 consecutive entries (8
// lanes x 16 bytes) and all active columns, dealt to NTH / 8 column groups whose sums meet in LDS in group order (the scheme of
// k_gemv_n1). Epilogue: dx on the free variables only (the fixed ones keep the move of their bound), and the two products the
// homotopy step integrates, A'dl_C and H dx, are never formed -- only their difference A'dl_C - H dx = (gN - g) - dl_B enters
// the gradient the drift correction derives (g = A'y + y_B - H x), so ATdy := 0 and Hdx := dl_B - (gN - g) carry it (both
// products are recomputed from the iterate every 8 changes as before).
template <int NTH>
__global__ void __launch_bounds__(NTH) k_ww_step(const double *__restrict__ WW, long long ldw, int nV, int nC, int nR,
                                                 const int *__restrict__ R, const double *__restrict__ dl, const double *__restrict__ pp,
                                                 const int *__restrict__ Sb, const double *__restrict__ dy, const double *__restrict__ gN,
                                                 const double *__restrict__ g, double *__restrict__ dx, double *__restrict__ dAx,
                                                 double *__restrict__ Hdx, double *__restrict__ ATdy) {
    constexpr int NG = NTH / 8;
    __shared__ double sh[NG][17];
    const int rl = threadIdx.x & 7, cg = threadIdx.x >> 3;
    const int n = nV + nC, r = (blockIdx.x * 8 + rl) * 2;
    double a0 = 0.0, a1 = 0.0;
    if (r + 1 < n) {
        double2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0};
        int c = cg;
        for (; c + 3 * NG < nR; c += 4 * NG) {
            const double w0 = dl[c], w1 = dl[c + NG], w2 = dl[c + 2 * NG], w3 = dl[c + 3 * NG];
            const double2 m0 = *reinterpret_cast<const double2 *>(WW + (long long)R[c] * ldw + r);
            const double2 m1 = *reinterpret_cast<const double2 *>(WW + (long long)R[c + NG] * ldw + r);
            const double2 m2 = *reinterpret_cast<const double2 *>(WW + (long long)R[c + 2 * NG] * ldw + r);
            const double2 m3 = *reinterpret_cast<const double2 *>(WW + (long long)R[c + 3 * NG] * ldw + r);
            s0.x += m0.x * w0; s0.y += m0.y * w0; s1.x += m1.x * w1; s1.y += m1.y * w1;
            s2.x += m2.x * w2; s2.y += m2.y * w2; s3.x += m3.x * w3; s3.y += m3.y * w3;
        }
        for (; c < nR; c += NG) {
            const double w0 = dl[c];
            const double2 m0 = *reinterpret_cast<const double2 *>(WW + (long long)R[c] * ldw + r);
            s0.x += m0.x * w0; s0.y += m0.y * w0;
        }
        a0 = (s0.x + s1.x) + (s2.x + s3.x);
        a1 = (s0.y + s1.y) + (s2.y + s3.y);
    } else if (r < n) {
        for (int c = cg; c < nR; c += NG) a0 += WW[(long long)R[c] * ldw + r] * dl[c];
    }
    sh[cg][2 * rl] = a0; sh[cg][2 * rl + 1] = a1;
    __syncthreads();
    if ((int)threadIdx.x < 16) {
        const int i = blockIdx.x * 16 + threadIdx.x;
        if (i < n) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int q = 0; q < NG; q += 4) { t0 += sh[q][threadIdx.x]; t1 += sh[q + 1][threadIdx.x]; t2 += sh[q + 2][threadIdx.x]; t3 += sh[q + 3][threadIdx.x]; }
            const double v = ((t0 + t1) + (t2 + t3)) - pp[i];
            if (i < nV) {
                const bool fixed = Sb[i] != 0;
                if (!fixed) dx[i] = v;
                Hdx[i] = (fixed ? dy[i] : 0.0) - (gN[i] - g[i]);
                ATdy[i] = 0.0;
            } else dAx[i - nV] = v;
        }
    }
}
// S = C H^-1 C' of a guessed working set is a sub-matrix of the tableau: G[i][j] = WW[R[i]][R[j]] (upper triangle)
__global__ void k_ww_gather_S(int n, const int *__restrict__ R, const double *__restrict__ WW, long long ldw, double *__restrict__ G, long long ldg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < n && i <= j) G[i + (long long)j * ldg] = WW[R[i] + (long long)R[j] * ldw];
}

// ---- one step of iterative refinement on the final KKT system (rs_refine) --------------------------------------------------------
// rg = H x + g - A'y_C on the free variables (the stationarity residual there; on a fixed variable it DEFINES the bound's multiplier)
__global__ void k_rs_refine_rg(int nV, const int *__restrict__ Sb, const double *__restrict__ Hx, const double *__restrict__ g,
                               const double *__restrict__ ATy, double *__restrict__ rg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV) rg[i] = Sb[i] == 0 ? (Hx[i] + g[i]) - ATy[i] : 0.0;
}
// right-hand side over the active rows: what the row lacks + (C H^-1 rg)   (x sits exactly on its active bounds: nothing lacks there)
__global__ void k_rs_refine_rhs(int nR, const int *__restrict__ R, int nV, const int *__restrict__ Sall, const double *__restrict__ lbA,
                                const double *__restrict__ ubA, const double *__restrict__ Ax, const double *__restrict__ t,
                                const double *__restrict__ At, double *__restrict__ rhs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nR) return;
    const int id = R[j];
    if (id < nV) rhs[j] = t[id];
    else { const int r = id - nV; rhs[j] = ((Sall[id] == -1 ? lbA[r] : ubA[r]) - Ax[r]) + At[r]; }
}
// q = A'dl_C + dl_B - rg  (the product with H^-1 gives the correction of x)
__global__ void k_rs_refine_q(int nV, const int *__restrict__ Sb, const double *__restrict__ ATdy, const double *__restrict__ dy,
                              const double *__restrict__ rg, double *__restrict__ q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV) q[i] = (ATdy[i] + (Sb[i] != 0 ? dy[i] : 0.0)) - rg[i];
}
// x += dx on the free variables, y_C += dl_C
__global__ void k_rs_refine_apply(int nV, int nC, const int *__restrict__ Sb, const int *__restrict__ Sc, const double *__restrict__ dx,
                                  const double *__restrict__ dy, double *__restrict__ x, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV && Sb[i] == 0) x[i] += dx[i];
    if (i < nC && Sc[i] != 0) y[nV + i] += dy[nV + i];
}
// the multipliers of the fixed variables from stationarity with the corrected x and y_C
__global__ void k_rs_refine_yb(int nV, const int *__restrict__ Sb, const double *__restrict__ Hx, const double *__restrict__ g,
                               const double *__restrict__ ATy, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV && Sb[i] != 0) y[i] = (Hx[i] + g[i]) - ATy[i];
}

// ---- set-up ---------------------------------------------------------------------------------------------------------------------
// Cd[:, j] = row R[j] of C as a dense column (columns zero-filled beforehand)
__global__ void k_rs_build_C(int nV, const int *__restrict__ R, const int *__restrict__ rp, const int *__restrict__ ci,
                             const double *__restrict__ rv, double *__restrict__ Cd, long long ld) {
    const int id = R[blockIdx.x];
    double *col = Cd + (long long)blockIdx.x * ld;
    if (id < nV) { if (threadIdx.x == 0) col[id] = 1.0; return; }
    const int r = id - nV;
    for (int k = rp[r] + threadIdx.x; k < rp[r + 1]; k += blockDim.x) col[ci[k]] = rv[k];
}
// Sinv of the working set "every variable fixed, no constraint": S = H^-1, so Sinv = H itself (upper triangle, dense; zero-filled
// beforehand). One thread per stored entry of the CSC.
__global__ void k_rs_sinv_from_H(int nV, const int *__restrict__ jc, const int *__restrict__ ir, const double *__restrict__ val,
                                 double hreg, double *__restrict__ Sinv, long long ld) {
    const int c = blockIdx.x;
    for (int k = jc[c] + threadIdx.x; k < jc[c + 1]; k += blockDim.x) {
        const int r = ir[k];
        if (r <= c) Sinv[(long long)c * ld + r] = val[k] + (r == c ? hreg : 0.0);
    }
}
__global__ void k_rs_iota(int n, int *__restrict__ R, int *__restrict__ pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { R[i] = i; pos[i] = i; }
}

// qp_rs_path.h -- host side of the GENERAL range-space path of the HBM-resident engine (RsqpLargeEngine::Impl::rsh). Included by
// qp_large.hip behind the definition of Impl; formulation and kernels: qp_rs_kernels.h, DESIGN 4.5.
//
// Which solves take it: H present, symmetric and positive definite, not the diagonal special case of DESIGN 4.4 (that one keeps
// its own, cheaper treatment of bounds), and an H^-1 operator can be built -- banded (half bandwidth <= 2, nV <= 16 384) or
// dense (nV <= 16 384). Everything around the factors -- homotopy, ratio tests, tie breaks, drift correction, exchange rule, host
// protocol -- is the code of qp_large.hip, so the working-set sequences are those of the null-space path.
#pragma once

using Impl = RsqpLargeEngine::Impl;

// ---- the H^-1 operator -------------------------------------------------------------------------------------------------------------
// banded: LDL' on the host (O(n) work, the values come over once per set-up), homogeneous solutions per chunk, nine arrays up
int Impl::rs_build_band(const std::vector<double> &hv, bool *ok) {
    *ok = false;
    const int n = nV;
    std::vector<double> d0(n, 0.0), e1(n, 0.0), e2(n, 0.0);      // H[i][i], H[i][i-1], H[i][i-2]
    for (int c = 0; c < n; c++)
        for (int k = M.h_Hjc[c]; k < M.h_Hjc[c + 1]; k++) {
            const int r = M.h_Hir[k];
            if (r == c) d0[c] += hv[k];
            else if (r == c + 1) e1[r] += hv[k];
            else if (r == c + 2) e2[r] += hv[k];
        }
    for (int i = 0; i < n; i++) d0[i] += M.hreg;
    std::vector<double> d(n), l1(n, 0.0), l2(n, 0.0);
    for (int i = 0; i < n; i++) {
        if (i >= 2) l2[i] = e2[i] / d[i - 2];
        if (i >= 1) l1[i] = (e1[i] - (i >= 2 ? l2[i] * d[i - 2] * l1[i - 1] : 0.0)) / d[i - 1];
        double di = d0[i];
        if (i >= 1) di -= l1[i] * l1[i] * d[i - 1];
        if (i >= 2) di -= l2[i] * l2[i] * d[i - 2];
        if (!(di > 1e-12 * std::fabs(d0[i])) || !(di > 0.0)) return RET_OK;      // not positive definite (enough): not this path
        d[i] = di;
    }
    // (a factor whose homogeneous solutions grow along the rows -- H far from diagonal dominance -- would lose digits in the scan
    //  of chunk maps: such a Hessian takes the dense operator). Growth of the two homogeneous solutions over windows of 64 rows:
    {
        double growth = 0.0;
        for (int s0 = 0; s0 < n; s0 += 64) {
            double a1 = 1.0, a2 = 0.0, b1 = 0.0, b2 = 1.0;
            for (int i = s0; i < std::min(s0 + 64, n); i++) {
                const double na = -l1[i] * a1 - l2[i] * a2, nb = -l1[i] * b1 - l2[i] * b2;
                a2 = a1; a1 = na; b2 = b1; b1 = nb;
                growth = std::max(growth, std::max(std::fabs(na), std::fabs(nb)));
            }
        }
        if (!(growth <= 1e3)) return RET_OK;
    }
    const int T = BAND_TH, c = (n + T - 1) / T;
    const size_t nagg = 2 * BAND_MW * 8;
    std::vector<double> buf((size_t)T * (2 * c + 3) + (size_t)n + nagg, 0.0);
    double *m1i = buf.data(), *m2i = m1i + (size_t)T * (c + 1), *sinv = m2i + (size_t)T * (c + 2);
    std::vector<double> sq(n);
    for (int i = 0; i < n; i++) { sq[i] = std::sqrt(d[i]); sinv[i] = 1.0 / sq[i]; }
    for (int t = 0; t < T; t++) {
        for (int k = 0; k <= c + 1; k++) {
            const int i = t * c + k;
            if (k <= c) m1i[(size_t)k * T + t] = (i < n && i >= 1) ? l1[i] * sq[i - 1] / sq[i] : 0.0;
            m2i[(size_t)k * T + t] = (i < n && i >= 2) ? l2[i] * sq[i - 2] / sq[i] : 0.0;
        }
    }
    if (!band_buf) LCHK(hipMalloc(reinterpret_cast<void **>(&band_buf), sizeof(double) * buf.size()));
    LCHK(hipMemcpy(band_buf, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice));
    band.n = n; band.c = c;
    band.m1i = band_buf; band.m2i = band.m1i + (size_t)T * (c + 1); band.sinv = band.m2i + (size_t)T * (c + 2);
    band.agg = band_buf + (size_t)T * (2 * c + 3) + (size_t)n;
    band.err = dflag + 1;
    band_seq = 0.0;
    LCHK(hipMemsetAsync(dflag + 1, 0, sizeof(int), st));
    if (!band_attr_set) {      // (per handle: a function attribute belongs to the device the handle lives on)
        LCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_apply<4>), hipFuncAttributeMaxDynamicSharedMemorySize, BAND_MAX_N * 8));
        LCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_apply<8>), hipFuncAttributeMaxDynamicSharedMemorySize, BAND_MAX_N * 8));
        LCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_apply<10>), hipFuncAttributeMaxDynamicSharedMemorySize, BAND_MAX_N * 8));
        LCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_apply<12>), hipFuncAttributeMaxDynamicSharedMemorySize, BAND_MAX_N * 8));
        LCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_apply<16>), hipFuncAttributeMaxDynamicSharedMemorySize, BAND_MAX_N * 8));
        band_attr_set = true;
    }
    // the explicit inverse as well (in Z: n^2 doubles, 0.8 GB at n = 10 000): H^-1 c' of an incoming row -- a unit vector or a few
    // entries -- is then a combination of a few of its columns (k_rs_hinv_row, every lane busy) instead of a banded product whose
    // two scans and two cross-workgroup hand-offs take 20 us whatever the vector holds. One banded product per column, all at once.
    band_hinv = false;
    if (rs_band_columns) {
        LCHK(hipMemsetAsync(Z, 0, sizeof(double) * (size_t)ld * nV, st));
        hipLaunchKernelGGL(k_rs_unit_diag, g1(nV), dim3(NT), 0, st, nV, Z, ld);
        band_launch(nV, Z, nullptr, Z, ld, false);
        band_hinv = true;
    }
    *ok = true;
    return RET_OK;
}
// dense: Cholesky, triangular inverse, U^-1 U^-T on the matrix cores (dense_la.hip), both triangles kept (columns are gathered)
int Impl::rs_build_dense(bool *ok) {
    *ok = false;
    if (ensure_big() != RET_OK) return RET_SETUP_FAILED;
    double *G = big, *Ui = big + (size_t)ld * ld;
    LCHK(hipMemsetAsync(G, 0, sizeof(double) * (size_t)ld * nV, st));
    hipLaunchKernelGGL(k_rs_sinv_from_H, dim3(nV), dim3(64), 0, st, nV, M.Hjc, M.Hir, M.Hval, M.hreg, G, ld);
    LCHK(hipMemsetAsync(dw.flag, 0, sizeof(int) * 4, st));
    LCHK(rsqp_dpotrf_upper(nV, G, ld, 1e-12, 0.0, &dw, st));
    LCHK(hipMemcpyAsync(h_pinned_i, dw.flag, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    LCHK(hipStreamSynchronize(st));
    if (h_pinned_i[1] != 0) return RET_OK;
    LCHK(rsqp_dtrtri_upper(nV, G, ld, Ui, ld, &dw, st));
    LCHK(rsqp_dtrmmt_upper(nV, 1.0, Ui, ld, Z, ld, st));
    LCHK(rsqp_mirror_upper(nV, Z, ld, st));
    *ok = true;
    return RET_OK;
}
// the static tableau of a small dense problem: WW = [I; A] H^-1 [I A'], (nV + nC)^2, three GEMMs once per Hessian
int Impl::rs_build_ww() {
    const int n = nV + nC;
    ldw = pad16(n);
    if (!rs_WW) LCHK(hipMalloc(reinterpret_cast<void **>(&rs_WW), sizeof(double) * (size_t)ldw * n));
    LCHK(hipMemcpy2DAsync(rs_WW, sizeof(double) * ldw, Z, sizeof(double) * ld, sizeof(double) * nV, nV, hipMemcpyDeviceToDevice, st));
    if (nC <= 0) return RET_OK;
    const double *dA = M.denseA;
    if (!dA) {
        if (!rs_dA) LCHK(hipMalloc(reinterpret_cast<void **>(&rs_dA), sizeof(double) * (size_t)nC * nV));
        LCHK(rsqp_launch_densify(nC, nV, M.Ajc, M.Air, M.Aval, rs_dA, st));
        dA = rs_dA;
    }
    double *TR = rs_WW + (size_t)nV * ldw, *BL = rs_WW + nV, *BR = rs_WW + nV + (size_t)nV * ldw;
    LCHK(rsqp_dgemm(false, true, nV, nC, nV, 1.0, Z, ld, dA, nC, 0.0, TR, ldw, st));        // H^-1 A'
    LCHK(rsqp_dgemm(false, false, nC, nV, nV, 1.0, dA, nC, Z, ld, 0.0, BL, ldw, st));       // A H^-1
    LCHK(rsqp_dgemm(false, false, nC, nC, nV, 1.0, dA, nC, TR, ldw, 0.0, BR, ldw, st));     // A H^-1 A'
    chk("rs_build_ww");
    return RET_OK;
}
int Impl::rs_prepare(bool *ok) {
    *ok = false;
    if (!rsh_enabled || !M.haveH || !M.h_Hjc || !M.h_Hir || M.Hnnz <= 0 || nV > BAND_MAX_N) return RET_OK;
    // symmetry (pattern and values) and band width: on the device, two words back
    LCHK(hipMemsetAsync(dflag + 2, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(k_rs_sym_check, dim3(nV), dim3(NT), 0, st, nV, M.Hjc, M.Hir, M.Hval, dflag + 2);
    LCHK(hipMemcpyAsync(h_pinned_i, dflag + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    LCHK(hipStreamSynchronize(st));
    const int hb = h_pinned_i[0];
    if (h_pinned_i[1] != 0) return RET_OK;
    std::vector<double> hv;
    if (hb <= 2 && !rs_force_dense) {      // the banded factor is built on the host: its few values come over
        hv.resize(M.Hnnz);
        LCHK(hipMemcpyAsync(hv.data(), M.Hval, sizeof(double) * M.Hnnz, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
    }
    bool built = false;
    if (hb <= 2 && !rs_force_dense) { if (rs_build_band(hv, &built) != RET_OK) return RET_SETUP_FAILED; if (built) rs_kind = 1; }
    if (!built) {
        if (rs_build_dense(&built) != RET_OK) return RET_SETUP_FAILED;
        if (built) rs_kind = 2;
        if (built && !rs_no_ww && nV + nC <= WW_MAX) { if (rs_build_ww() != RET_OK) return RET_SETUP_FAILED; rs_kind = 3; }
    }
    *ok = built;
    return RET_OK;
}
void Impl::band_launch(int ncols, const double *in, const double *sub, double *out, long long ldc, bool qmode) {
    BandQ q{};
    if (qmode) { q.Sb = Sb; q.ATdy = ATdy; q.dy = dy; q.gN = gN; q.g = g; q.Hdx = Hdx; }
    const int c = band.c;
    if (ncols == 1 && band_mw) {       // one vector: 16 single-wave workgroups
        const double seq = (band_seq += 1.0);
#define RS_MW(CC) hipLaunchKernelGGL(k_band_apply_mw<CC>, dim3(BAND_MW), dim3(64), 0, st, band, in, sub, out, q, seq)
        if (c <= 4) RS_MW(4); else if (c <= 8) RS_MW(8); else if (c <= 10) RS_MW(10); else if (c <= 12) RS_MW(12); else RS_MW(16);
#undef RS_MW
        return;
    }
    const size_t lds = sizeof(double) * (size_t)nV;
#define RS_1W(CC) hipLaunchKernelGGL(k_band_apply<CC>, dim3(ncols), dim3(BAND_TH), lds, st, band, in, sub, out, ldc, q)
    if (c <= 4) RS_1W(4); else if (c <= 8) RS_1W(8); else if (c <= 10) RS_1W(10); else if (c <= 12) RS_1W(12); else RS_1W(16);
#undef RS_1W
}
// out = H^-1 (in - sub); fix_dx: the step direction's product -- in = q - (gN - g) formed on the way (into Hdx), out = dx written on
// the free variables only (in / sub are ignored)
void Impl::rs_hinv_apply(const double *in, const double *sub, double *out, bool fix_dx) {
    pbegin();
    if (rs_kind == 1) {
        band_launch(1, in, sub, out, 0LL, fix_dx);
        pend(9, 40.0 * nV);
    } else if (fix_dx) {
        hipLaunchKernelGGL(k_rs_q, g1(nV), dim3(NT), 0, st, nV, Sb, ATdy, dy, gN, g, Hdx);
        gemv_n(Z, ld, nV, nV, Hdx, 1.0, 0.0, nullptr, w4, Sb, out);       // (merge epilogue: out = the product on the free variables only)
    } else {
        const double *src = in;
        if (sub) { hipLaunchKernelGGL(k_rs_diff, g1(nV), dim3(NT), 0, st, nV, in, sub, wz3); src = wz3; }
        gemv_n(Z, ld, nV, nV, src, 1.0, 0.0, nullptr, out);
    }
    chk("rs_hinv_apply");
}

// ---- Sinv ------------------------------------------------------------------------------------------------------------------------
void Impl::rs_rank1(int n, const double *v, int slot, double cs) {
    if (n <= 0) return;
    pbegin();
    hipLaunchKernelGGL((k_sym_tile<true, false>), dim3(sym_tiles(n)), dim3(256), 0, st, Wz, ld, n, v, scal, slot, cs, (const double *)nullptr,
                       (double *)nullptr, (double *)nullptr);
    pend(2, 8.0 * n * (double)n);
}
void Impl::rs_flush() {
    if (!rsh) return;
    if (lz_n > 0 && nR > 0) {      // all pending terms in one pass over the triangle
        pbegin();
        hipLaunchKernelGGL(k_sym_tile_lz, dim3(sym_tiles(nR)), dim3(256), 0, st, Wz, ld, nR, lz_p());
        pend(2, 8.0 * nR * (double)nR);
    }
    lz_n = 0;
    if (!pendR.on) return;
    pendR.on = false;
    rs_rank1(pendR.n, rs_ps_u, S_KEEP_S, 1.0);
}
// out = Sinv w. Lazy form: one read of the triangle, the pending terms in the reduction; otherwise the deferred rank-1 part of the
// last bordering is applied on the way
void Impl::rs_sinv_times(const double *wv, double *out, bool scatter_dy) {
    if (nR <= 0) return;
    const int nt = (nR + SYT - 1) / SYT;
    double *P1 = wz_part, *P2 = wz_part + (size_t)nt * nR;
    if (lz_enabled) {
        if (lz_n > 0) hipLaunchKernelGGL(k_lz_dots, dim3(lz_n), dim3(NT), 0, st, lz_p(), nR, wv, lz_d);
        pbegin();
        hipLaunchKernelGGL((k_sym_tile<false, true>), dim3(sym_tiles(nR)), dim3(256), 0, st, Wz, ld, nR, (const double *)nullptr, scal, 0, 0.0, wv, P1, P2);
        pend(0, 4.0 * nR * (double)nR);
        hipLaunchKernelGGL(k_sym_reduce_lz, g1(nR), dim3(NT), 0, st, nR, nt, P1, P2, out, scatter_dy ? R : (const int *)nullptr,
                           scatter_dy ? dy : (double *)nullptr, lz_p(), lz_d);
        chk("rs lazy product");
        return;
    }
    pbegin();
    if (pendR.on && nR == pendR.n + 1) {
        pendR.on = false;
        hipLaunchKernelGGL((k_sym_tile<true, true>), dim3(sym_tiles(nR)), dim3(256), 0, st, Wz, ld, nR, rs_ps_u, scal, S_KEEP_S, 1.0, wv, P1, P2);
        pend(8, 8.0 * nR * (double)nR);
    } else {
        rs_flush();
        hipLaunchKernelGGL((k_sym_tile<false, true>), dim3(sym_tiles(nR)), dim3(256), 0, st, Wz, ld, nR, (const double *)nullptr, scal, 0, 0.0, wv, P1, P2);
        pend(0, 4.0 * nR * (double)nR);
    }
    hipLaunchKernelGGL(k_sym_reduce, g1(nR), dim3(NT), 0, st, nR, nt, P1, P2, out, scatter_dy ? R : (const int *)nullptr, scatter_dy ? dy : (double *)nullptr);
    chk("rs sym product");
}
void Impl::rs_count(int id, int delta) {       // delta +1: the row joined, -1: it left
    if (id < nV) nFR -= delta; else nAC += delta;
    nZ = nFR - nAC;
}

// ---- a row joins / leaves ---------------------------------------------------------------------------------------------------------
// products of the incoming row with the working set: c in w1 (all variables), w = H^-1 c' in w5, cv = C w in ra1, u = Sinv cv in
// ra2; |c_FR|^2, the pivot s and c w published (stage 1 of the independence test)
void Impl::rs_products(int id) {
    rs_cur_id = id;
    if (rs_kind == 3) {      // everything about the row is a column of the tableau
        hipLaunchKernelGGL(k_ww_cv, g1(std::max(nR, 1)), dim3(NT), 0, st, nR, R, rs_WW, ldw, id, ra1, scal, S_WW_AD);
        rs_sinv_times(ra1, ra2);
        hipLaunchKernelGGL(k_ww_li_publish, dim3(1), dim3(NT), 0, st, nV, Sb, id, M.denseAT, M.Arp, M.Aci, M.Arv, nR, ra1, ra2, scal, S_WW_AD, d_ctl, next_seq());
        return;
    }
    hipLaunchKernelGGL(k_rs_row, dim3(1), dim3(NT), 0, st, nV, id, M.Arp, M.Aci, M.Arv, M.denseAT, w1);
    if ((rs_kind == 2 || (rs_kind == 1 && band_hinv)) && !M.denseAT) hipLaunchKernelGGL(k_rs_hinv_row, g1(nV), dim3(NT), 0, st, nV, id, M.Arp, M.Aci, M.Arv, Z, ld, w5);
    else rs_hinv_apply(w1, nullptr, w5, false);
    A_times(w5, c3);
    if (nR > 0) hipLaunchKernelGGL(k_rs_gather, g1(nR), dim3(NT), 0, st, nR, R, nV, w5, c3, ra1);
    rs_sinv_times(ra1, ra2);
    hipLaunchKernelGGL(k_rs_li_publish, dim3(1), dim3(1024), 0, st, nV, Sb, w1, w5, (const double *)nullptr, nR, ra1, ra2, scal, d_ctl, next_seq());
}
// stage 2: the first-order residual (also what an exchange needs: c1 = u by constraint, w2 = A_AC'u_C)
void Impl::rs_residual() {
    fill(c1, std::max(nC, 1), 0.0);
    if (nR > 0) hipLaunchKernelGGL(k_rs_scatter_c, g1(nR), dim3(NT), 0, st, nR, R, nV, ra2, c1);
    AT_times(c1, w2);
    hipLaunchKernelGGL(k_rs_li_publish, dim3(1), dim3(1024), 0, st, nV, Sb, w1, w5, w2, nR, ra1, ra2, scal, d_ctl, next_seq());
}
int Impl::rs_li_decision(bool *li) {
    if (wait_ctl() != RET_OK) return wait_failed();
    double a2n = h_ctl[2], sp = h_ctl[4];
    const double ad = h_ctl[5];
    if (nR < nV && a2n > 0.0 && sp > 1e-6 * ad) { *li = true; return RET_OK; }
    if (rs_kind == 3) {      // the residual test wants the row and H^-1 of it as vectors: the row from A, the other from the tableau
        hipLaunchKernelGGL(k_rs_row, dim3(1), dim3(NT), 0, st, nV, rs_cur_id, M.Arp, M.Aci, M.Arv, M.denseAT, w1);
        copy(rs_WW + (size_t)rs_cur_id * ldw, w5, nV);
    }
    rs_residual();
    if (wait_ctl() != RET_OK) return wait_failed();
    a2n = h_ctl[2]; sp = h_ctl[4];
    const double r2 = h_ctl[3];
    *li = nR < nV && a2n > 0.0 && std::sqrt(r2) > RSQP_EPS_LI * std::sqrt(a2n) && sp > 0.0;
    return RET_OK;
}
// Sinv <- [[Sinv + u u'/s, -u/s], [-u'/s, 1/s]]  (u in ra2, 1/s in scal[8]); the rank-1 part rides on the next product
void Impl::rs_add_row(int id, int side, int yidx, double yval) {
    if (lz_enabled) {
        if (lz_n >= LZK) rs_flush();
        // (nR = 0: no rank-1 part; the term is recorded with an empty vector all the same)
        hipLaunchKernelGGL(k_lz_border, g1(nR + 1), dim3(NT), 0, st, Wz, ld, nR, ra2, scal, R, posR, Sall, id, side, y, yidx, yval, lz_vec, lz_stride, lz_n, lz_c);
        lz_n++;
    } else {
        rs_flush();
        const bool defer = nR > 0;
        if (defer) { pendR.on = true; pendR.n = nR; }
        hipLaunchKernelGGL(k_dual_border_sym, g1(nR + 1), dim3(NT), 0, st, Wz, ld, nR, ra2, scal, R, posR, Sall, id, side, y, yidx, yval,
                           defer ? rs_ps_u : (double *)nullptr, S_KEEP_S);
    }
    hR[nR] = id; hposR[id] = nR;
    if (id < nV) hSb[id] = side; else hSc[id - nV] = side;
    nR++;
    rs_count(id, +1);
}
void Impl::rs_remove_row(int k, bool carry) {
    const int id = hR[k];
    if (lz_enabled) {
        if (lz_n >= LZK) rs_flush();
        hipLaunchKernelGGL(k_lz_colcoef, g1(nR), dim3(NT), 0, st, Wz, ld, nR, k, lz_p(), ra3, scal);
        if (carry) hipLaunchKernelGGL(k_dual_carry_remove, dim3(1), dim3(NT), 0, st, nR, k, 1.0 - last_tau, ra3, rs_dl);
        hipLaunchKernelGGL(k_lz_remove_fix, g1(std::max(nR, 1)), dim3(NT), 0, st, nR, k, ra3, scal, lz_vec, lz_stride, lz_n, lz_c);
        lz_n++;
    } else {
        rs_flush();
        hipLaunchKernelGGL(k_dual_colcoef_sym, g1(nR), dim3(NT), 0, st, Wz, ld, nR, k, ra3, scal);
        if (carry) hipLaunchKernelGGL(k_dual_carry_remove, dim3(1), dim3(NT), 0, st, nR, k, 1.0 - last_tau, ra3, rs_dl);
        rs_rank1(nR, ra3, 9, 1.0);
    }
    hipLaunchKernelGGL(k_dual_move_last_sym, g1(std::max(nR - 1, 1)), dim3(NT), 0, st, Wz, ld, nR, k, R, posR, Sall, id, y, id);
    if (k != nR - 1) { hR[k] = hR[nR - 1]; hposR[hR[k]] = k; }
    hposR[id] = -1;
    if (id < nV) hSb[id] = 0; else hSc[id - nV] = 0;
    nR--;
    rs_count(id, -1);
}
int Impl::rs_change_active_set(int kind, int idx, int side) {
    carry_pending = carry_ready = false;
    const int id = (kind == 1 || kind == 3) ? nV + idx : idx;
    if (kind == 1 || kind == 2) {
        // a row leaves: the multiplier step is carried (rs_dl transformed here, with the column of Sinv as it is before the update)
        const bool will_carry = rs_carry_enabled && carry_valid && carried < CARRY_REFRESH && nR > 1;
        rs_remove_row(hposR[id], will_carry);
        carry_ready = will_carry;
        return RET_OK;
    }
    double ynew = 0.0;
    bool li = false, exchanged = false;
    rs_products(id);
    if (rs_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
    if (!li) {
        int pkind = 0, pidx = -1;
        exchanged = true;
        hipLaunchKernelGGL(k_rs_row, dim3(1), dim3(NT), 0, st, nV, id, M.Arp, M.Aci, M.Arv, M.denseAT, w4);
        const int rc = ensure_LI(side, &ynew, &pkind, &pidx);
        if (rc != RET_OK) return rc;
        const int pid_ = pkind == 1 ? nV + pidx : pidx;
        rs_remove_row(hposR[pid_]);
        rs_products(id);
        if (rs_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
        if (!li && !(h_ctl[4] > 1e-14 * h_ctl[5])) return RET_SETUP_FAILED;
    }
    rs_add_row(id, side, id, ynew);
    // a plain addition (no exchange before it: cv in ra1, u in ra2, 1 / s in scal[8] are those of the bordering)
    carry_pending = rs_carry_enabled && ynew == 0.0 && li && carry_valid && carried < CARRY_REFRESH && !exchanged;
    rs_carry_id = id;
    return RET_OK;
}

// ---- step direction ----------------------------------------------------------------------------------------------------------------
void Impl::rs_refresh_p() {
    if (rs_kind == 3) {      // [p; A p] = WW[:, variables] (gN - g)
        hipLaunchKernelGGL(k_rs_diff, g1(nV), dim3(NT), 0, st, nV, gN, g, wz3);
        gemv_n(rs_WW, ldw, nV + nC, nV, wz3, 1.0, 0.0, nullptr, rs_p);
        return;
    }
    rs_hinv_apply(gN, g, rs_p, false);
    A_times(rs_p, rs_Ap);
}
void Impl::rs_step_direction() {
    if (!dx_ready) hipLaunchKernelGGL(k_dx_fixed_zero_dy, g1(nV + nC), dim3(NT), 0, st, nV, nC, Sb, lb, ub, lbN, ubN, dx, dy);
    dx_ready = false;
    if (nR > 0) {
        if (carry_ready && carry_valid) {                     // (transformed by rs_remove_row already)
            hipLaunchKernelGGL(k_rs_scatter, g1(nR), dim3(NT), 0, st, nR, R, rs_dl, dy);
            carried++; stat_carried++;
        } else if (carry_pending && carry_valid) {
            hipLaunchKernelGGL(k_rs_carry_add, dim3(1), dim3(1024), 0, st, nR - 1, 1.0 - last_tau, ra1, ra2, rs_dl, scal, rs_carry_id, nV, Sall, lb, ub,
                               lbN, ubN, lbA, ubA, lbAN, ubAN, rs_p, rs_Ap, R, dy);
            carried++; stat_carried++;
        } else {
            hipLaunchKernelGGL(k_rs_rhs, g1(nR), dim3(NT), 0, st, nR, R, nV, Sall, lb, ub, lbN, ubN, lbA, ubA, lbAN, ubAN, rs_p, rs_Ap, ra4);
            rs_sinv_times(ra4, rs_dl, true);
            carried = 0;
        }
    }
    if (rs_kind == 3) {
        pbegin();
        hipLaunchKernelGGL((k_ww_step<512>), dim3((nV + nC + 15) / 16), dim3(512), 0, st, rs_WW, ldw, nV, nC, nR, R, rs_dl, rs_p, Sb, dy, gN, g, dx, dAx, Hdx, ATdy);
        pend(0, 8.0 * (nV + nC) * (double)nR);
    } else {
        AT_times(dy + nV, ATdy);
        rs_hinv_apply(nullptr, nullptr, dx, true);
        A_times(dx, dAx);
    }
    carry_pending = carry_ready = false;
    carry_valid = nR > 0;
    chk("rs_step_direction");
}

// One step of iterative refinement on the KKT system of the FINAL working set, residuals formed from the data: the explicit inverse
// of C H^-1 C' squares the conditioning of the active rows and is kept current by rank-1 formulas, so a long homotopy on a vertex
// solution leaves x / y with ~1e-9 relative error (the randomised large-engine check under tests/checks, a 64 x 212 member: 3e-9) where the orthogonal
// factors of the null-space path give 1e-13; one correction through the same Sinv brings it back to working precision. The tableau
// engines end a solve the same way. Leaves Ax, ATy, Hx exact for the corrected point.
void Impl::rs_refine() {
    rs_flush();
    hipLaunchKernelGGL(k_fix_x, g1(nV), dim3(NT), 0, st, nV, Sb, lb, ub, x);
    refresh_products();                                             // Ax, A'y_C, H x of the point the homotopy ended in
    hipLaunchKernelGGL(k_rs_refine_rg, g1(nV), dim3(NT), 0, st, nV, Sb, Hx, g, ATy, w1);
    rs_hinv_apply(w1, nullptr, w5, false);                          // t = H^-1 rg
    A_times(w5, c3);
    fill(dy, nV + nC, 0.0);
    if (nR > 0) {
        hipLaunchKernelGGL(k_rs_refine_rhs, g1(nR), dim3(NT), 0, st, nR, R, nV, Sall, lbA, ubA, Ax, w5, c3, ra4);
        rs_sinv_times(ra4, ra3, true);                              // (scatters into dy)
    }
    AT_times(dy + nV, ATdy);
    hipLaunchKernelGGL(k_rs_refine_q, g1(nV), dim3(NT), 0, st, nV, Sb, ATdy, dy, w1, w2);
    rs_hinv_apply(w2, nullptr, w4, false);
    hipLaunchKernelGGL(k_rs_refine_apply, g1(std::max(nV, nC)), dim3(NT), 0, st, nV, nC, Sb, Sc, w4, dy, x, y);
    refresh_products();
    hipLaunchKernelGGL(k_rs_refine_yb, g1(nV), dim3(NT), 0, st, nV, Sb, Hx, g, ATy, y);
    carry_valid = false;
    chk("rs_refine");
}

// ---- set-up -------------------------------------------------------------------------------------------------------------------------
// Sinv for a given list of rows by GEMM + Cholesky + inverse: Cd = C' dense, T = H^-1 Cd, S = Cd'T (upper tiles), U'U = S,
// Sinv = U^-1 U^-T. RET_FALLBACK: S is not positive definite to working precision (dependent rows in the guess)
int Impl::rs_setup_rows(const std::vector<int> &rows, const std::vector<int> &gb, const std::vector<int> &gc) {
    const int n = (int)rows.size();
    if (ensure_big() != RET_OK) return RET_SETUP_FAILED;
    if (!rs_G) LCHK(hipMalloc(reinterpret_cast<void **>(&rs_G), sizeof(double) * (size_t)ld * ld));
    double *Cd = big, *T = big + (size_t)ld * ld;
    LCHK(hipMemcpyAsync(R, rows.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
    if (rs_kind != 3) {
        LCHK(hipMemsetAsync(Cd, 0, sizeof(double) * (size_t)ld * n, st));
        hipLaunchKernelGGL(k_rs_build_C, dim3(n), dim3(64), 0, st, nV, R, M.Arp, M.Aci, M.Arv, Cd, ld);
    }
    if (!se0) { (void)hipEventCreate(&se0); (void)hipEventCreate(&se1); (void)hipEventCreate(&se2); }
    setup_stat = SetupStat();
    (void)hipEventRecord(se0, st);
    if (rs_kind == 3) hipLaunchKernelGGL(k_ww_gather_S, dim3(g1(n).x, n), dim3(NT), 0, st, n, R, rs_WW, ldw, rs_G, ld);      // S is a sub-matrix of the tableau
    else {
        if (rs_kind == 1) band_launch(n, Cd, nullptr, T, ld, false);
        else LCHK(rsqp_dgemm(false, false, nV, n, nV, 1.0, Z, ld, Cd, ld, 0.0, T, ld, st));
        LCHK(rsqp_dgemm_upper(true, false, n, nV, 1.0, Cd, ld, T, ld, 0.0, rs_G, ld, st));
    }
    (void)hipEventRecord(se1, st);
    LCHK(hipMemsetAsync(dw.flag, 0, sizeof(int) * 4, st));
    LCHK(rsqp_dpotrf_upper(n, rs_G, ld, 1e-10, RSQP_EPS_PD_ABS, &dw, st));
    LCHK(hipMemcpyAsync(h_pinned_i, dw.flag, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    LCHK(hipStreamSynchronize(st));
    if (h_pinned_i[1] != 0) return RET_FALLBACK;
    double *Ui = big;
    LCHK(rsqp_dtrtri_upper(n, rs_G, ld, Ui, ld, &dw, st));
    LCHK(rsqp_dtrmmt_upper(n, 1.0, Ui, ld, Wz, ld, st));
    (void)hipEventRecord(se2, st);
    {
        (void)hipEventSynchronize(se2);
        float ms1 = 0.f, ms2 = 0.f;
        (void)hipEventElapsedTime(&ms1, se0, se1); (void)hipEventElapsedTime(&ms2, se1, se2);
        const double dn = n, dm = nV;
        setup_stat.valid = 1; setup_stat.dual = 2; setup_stat.m = nV; setup_stat.n = n; setup_stat.nZ = nV - n;
        setup_stat.ms_tq = ms1; setup_stat.ms_wz = ms2;
        setup_stat.flops_tq = rs_kind == 3 ? 0.0 : dn * dn * dm + (rs_kind == 2 ? 2.0 * dm * dm * dn : 0.0); setup_stat.flops_wz = dn * dn * dn;
    }
    std::fill(hposR.begin(), hposR.end(), -1);
    std::fill(hSb.begin(), hSb.end(), 0); std::fill(hSc.begin(), hSc.end(), 0);
    nFR = nV; nAC = 0;
    for (int k = 0; k < n; k++) {
        const int id = rows[k];
        hR[k] = id; hposR[id] = k;
        if (id < nV) { hSb[id] = gb[id]; nFR--; } else { hSc[id - nV] = gc[id - nV]; nAC++; }
    }
    nR = n; nZ = nFR - nAC;
    LCHK(hipMemcpyAsync(posR, hposR.data(), sizeof(int) * (nV + nC), hipMemcpyHostToDevice, st));
    LCHK(hipMemcpyAsync(Sb, hSb.data(), sizeof(int) * nV, hipMemcpyHostToDevice, st));
    if (nC > 0) LCHK(hipMemcpyAsync(Sc, hSc.data(), sizeof(int) * nC, hipMemcpyHostToDevice, st));
    LCHK(hipStreamSynchronize(st));
    chk("rs_setup_rows");
    return RET_OK;
}
// the factors of a guessed working set (gb: bounds, gc: constraints; Sb already holds gb, Sc is zero, x is set)
int Impl::rs_setup(const std::vector<int> &gb, const std::vector<int> &gc) {
    pendR.on = false;
    lz_n = 0;
    A_times(x, Ax);
    std::vector<int> rows, cons;
    for (int v = 0; v < nV; v++) if (gb[v] != 0) rows.push_back(v);
    for (int r = 0; r < nC; r++) if (gc[r] != 0) cons.push_back(nV + r);
    const int nFX = (int)rows.size();
    nFR = nV - nFX; nAC = 0; nR = 0; nZ = nFR;
    std::fill(hposR.begin(), hposR.end(), -1);
    LCHK(hipMemsetAsync(posR, 0xff, sizeof(int) * (nV + nC), st));
    bool rows_done = false;
    if (nFX == nV) {
        // every variable fixed: S = H^-1, Sinv = H itself -- no factorisation (the cold start of a box-constrained QP)
        LCHK(hipMemsetAsync(Wz, 0, sizeof(double) * (size_t)ld * nV, st));
        hipLaunchKernelGGL(k_rs_sinv_from_H, dim3(nV), dim3(64), 0, st, nV, M.Hjc, M.Hir, M.Hval, M.hreg, Wz, ld);
        hipLaunchKernelGGL(k_rs_iota, g1(nV), dim3(NT), 0, st, nV, R, posR);
        for (int v = 0; v < nV; v++) { hR[v] = v; hposR[v] = v; }
        nR = nV;
        rows_done = true;
        // (no constraint can join while no variable is free: the guessed ones, if any, are dropped like dependent rows)
        cons.clear();
    }
    if (!rows_done && blocked_setup && nFX + (int)cons.size() >= BLOCKED_MIN && nFX + (int)cons.size() <= nV) {
        std::vector<int> all(rows);
        all.insert(all.end(), cons.begin(), cons.end());
        int rcb = rs_setup_rows(all, gb, gc);
        if (rcb == RET_OK) { rows_done = true; cons.clear(); }
        else if (rcb != RET_FALLBACK) return rcb;
        else if (nFX >= BLOCKED_MIN) {          // dependent constraints in the guess: the bounds blocked (H^-1_FX,FX is definite), the rest one by one
            rcb = rs_setup_rows(rows, gb, gc);
            if (rcb == RET_OK) rows_done = true;
            else if (rcb != RET_FALLBACK) return rcb;
        }
    }
    if (!rows_done) {
        // one bordering per row; Sb is cleared first so that the independence test sees the variables it has not fixed yet as free
        LCHK(hipMemsetAsync(Sb, 0, sizeof(int) * nV, st));
        std::fill(hSb.begin(), hSb.end(), 0);
        nFR = nV; nZ = nFR;
        for (int id : rows) {
            rs_products(id);
            bool li = false;
            if (rs_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
            if (!li) return RET_SETUP_FAILED;      // (a bound is never dependent on other bounds)
            rs_add_row(id, gb[id], -1, 0.0);
        }
    }
    // (yidx = -1: the multipliers of the guess, uploaded by solve(), are left alone)
    for (int id : cons) {
        rs_products(id);
        bool li = false;
        if (rs_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
        if (li) rs_add_row(id, gc[id - nV], -1, 0.0);
    }
    rs_flush();
    nZ = nFR - nAC;
    return RET_OK;
}

"""Experiment of the day for tools/attn_lab.py (timing only; several of these compute wrong results by design)."""
VARIANTS = {}
VARIANTS["r03"] = ["attention_r03.hip"]                      # round 3's kernels (git show <round-3 head>:... > tools/dbg/_lab/attention_r03.hip)
VARIANTS["r04a"] = ["attention_r04a.hip"]                    # first commit of round 4: rotated forward pipeline, tail through C

# ---- forward: the two workgroups of a CU run 27 vs 37 us (age arbitration): equalise them?
_RB = '''            __syncthreads();                                 // tile `it` visible; the other buffer is free for tile it + 1
            const T* sK = sbase + (it & 1) * fwd_stage_elems<T>();'''
VARIANTS["prio_flip1"] = [(_RB, "            if (it & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);\n" + _RB)]
VARIANTS["prio_flip2"] = [(_RB, "            if ((it >> 1) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);\n" + _RB)]
VARIANTS["prio_flip4"] = [(_RB, "            if ((it >> 2) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);\n" + _RB)]
VARIANTS["prio_young"] = [("    const int wg = xcd_remap(blockIdx.x, gridDim.x);\n    const int seg = grp_find(grp, wg);\n    const AttnArgs<T>& p = grp.seg[seg];\n    const int w = wg - grp.first[seg];\n    const int nqt = (p.N + FWD_QWG - 1) / FWD_QWG;",
                           "    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);\n    const int wg = xcd_remap(blockIdx.x, gridDim.x);\n    const int seg = grp_find(grp, wg);\n    const AttnArgs<T>& p = grp.seg[seg];\n    const int w = wg - grp.first[seg];\n    const int nqt = (p.N + FWD_QWG - 1) / FWD_QWG;")]

# LDS read ablations of the dK/dV loop (register copies instead)
VARIANTS["kv_notr"] = [('''                    for (int s = 0; s < 2; ++s) {
                        const int q16 = 32 * qb + 16 * s;
                        trf[s][0] = DT::tr_frag(sdO, q16, 0, lane);
                        trf[s][1] = DT::tr_frag(sdO, q16, 32, lane);
                        trf[s][2] = DT::tr_frag(sQ, q16, 0, lane);
                        trf[s][3] = DT::tr_frag(sQ, q16, 32, lane);
                    }''', '''                    for (int s = 0; s < 2; ++s)
                        for (int j = 0; j < 4; ++j) { trf[s][j] = kf[j]; if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(trf[s][j].v)); }''')]
VARIANTS["kv_norows"] = [('''                    for (int c = 0; c < 4; ++c) { qa[c] = DT::row_frag(sQ, row, c, half); oa[c] = DT::row_frag(sdO, row, c, half); }
                };''', '''                    for (int c = 0; c < 4; ++c) { qa[c] = kf[c]; oa[c] = vf[c]; if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(qa[c].v), "+v"(oa[c].v)); }
                };''')]
VARIANTS["kv_norowconst"] = [('''                    load_rowconst(0, cL0, cD0);
                    load_rows(0, qa0, oa0);
                    load_rowconst(1, cL1, cD1);''', '''                    cL0 = f32x16{0}; cD0 = f32x16{0}; cL1 = f32x16{0}; cD1 = f32x16{0};
                    asm volatile("" : "+v"(cL0), "+v"(cD0), "+v"(cL1), "+v"(cD1));
                    load_rows(0, qa0, oa0);''')]

# ---- dK/dV: what is the loop bound by?  (timing only, wrong results)
VARIANTS["kv_nograds"] = [('''                    for (int s = 0; s < 2; ++s) {
                        mma<T>(dv0, trf[s][0], pf[s]);
                        mma<T>(dv1, trf[s][1], pf[s]);
                        mma<T>(dk0, trf[s][2], dsf[s]);
                        mma<T>(dk1, trf[s][3], dsf[s]);
                    }''', '''                    for (int s = 0; s < 2; ++s) {
                        if constexpr (sizeof(T) == 2) asm volatile("" :: "v"(trf[s][0].v), "v"(trf[s][1].v), "v"(trf[s][2].v), "v"(trf[s][3].v), "v"(pf[s].v), "v"(dsf[s].v));
                    }''')]
VARIANTS["kv_noprobs"] = [('''                    for (int t = 0; t < 16; ++t) {
                        const float pv = fast_exp2(st[t]);
                        st[t] = pv;
                        ds[t] *= pv;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) { pf[s] = frag_from_acc<T>(st, s); dsf[s] = frag_from_acc<T>(ds, s); }''', '''                    for (int s = 0; s < 2; ++s) { pf[s] = kf[s]; dsf[s] = vf[s]; if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(pf[s].v), "+v"(dsf[s].v)); }
                    asm volatile("" :: "v"(st), "v"(ds));''')]
VARIANTS["kv_noscores"] = [('''                    st = mma_c<T>(qa[0], kf[0], cL);
#pragma unroll
                    for (int c = 1; c < 4; ++c) mma<T>(st, qa[c], kf[c]);
                    ds = mma_c<T>(oa[0], vf[0], cD);
#pragma unroll
                    for (int c = 1; c < 4; ++c) mma<T>(ds, oa[c], vf[c]);''', '''                    st = cL; ds = cD;
                    if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(st), "+v"(ds) : "v"(qa[0].v), "v"(qa[1].v), "v"(qa[2].v), "v"(qa[3].v), "v"(oa[0].v), "v"(oa[1].v), "v"(oa[2].v), "v"(oa[3].v));''')]
# pure skeletons of the dK/dV loop
VARIANTS["kv_mfma_only"] = VARIANTS["kv_noprobs"] + VARIANTS["kv_notr"] + VARIANTS["kv_norows"] + VARIANTS["kv_norowconst"] + [
    ('''            __syncthreads();                   // tile `it` visible; the other stage is free for tile it + 1
            if (it + 1 < nq) put(it + 1);
            if (it + 2 < nq) fetch(it + 2);
            if (kw0 < kvl) {                   // wave-uniform''', '''            if (kw0 < kvl) {                   // wave-uniform''')]
VARIANTS["kv_no_mfma"] = VARIANTS["kv_nograds"] + VARIANTS["kv_noscores"]

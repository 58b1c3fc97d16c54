"""Experiment of the day for tools/attn_lab.py (timing only; several of these compute wrong results by design)."""
VARIANTS = {}
VARIANTS["r03"] = ["attention_r03.hip"]                      # round 3's kernels (git show <round-3 head>:... > tools/dbg/_lab/attention_r03.hip)
VARIANTS["r04a"] = ["attention_r04a.hip"]                    # first commit of round 4: rotated forward pipeline, tail through C

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
VARIANTS["kv_occ1"] = [("template <typename T> size_t dkdv_smem() { return (size_t)2 * dkdv_stage_bytes<T>(); }",
                        "template <typename T> size_t dkdv_smem() { return (size_t)90 * 1024; }")]

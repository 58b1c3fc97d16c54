"""Experiment of the day for tools/attn_lab.py (timing only; several of these compute wrong results by design)."""
VARIANTS = {}
VARIANTS["r03"] = ["attention_r03.hip"]                      # round 3's kernels (git show HEAD:... > tools/dbg/_lab/attention_r03.hip)
VARIANTS["norot"] = [("constexpr bool FWD_ROTATED = true;", "constexpr bool FWD_ROTATED = false;")]
# ---- forward: how much do the LDS fragment reads cost?  (K / V fragments replaced by register copies of Q fragments)
VARIANTS["fwd_nolds"] = [
    ('''    auto load_k = [&](const T* sK, int kb, Frag<T> (&ka)[4]) {
        const T* arow = sK + (32 * kb + swz23(r)) * LDT + 8 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) ka[c] = frag_load<T>(arow + 16 * c);
    };''', '''    auto load_k = [&](const T* sK, int kb, Frag<T> (&ka)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { ka[c] = qf[kb][c]; asm volatile("" : "+v"(ka[c].v)); }
    };'''),
    ('''        for (int s = 0; s < 2; ++s) {
            vt[s][0] = frag_tr(sV, 32 * kb + 16 * s, 0, lane);
            vt[s][1] = frag_tr(sV, 32 * kb + 16 * s, 32, lane);
        }
    };''', '''        for (int s = 0; s < 2; ++s) {
            vt[s][0] = qf[s][0]; asm volatile("" : "+v"(vt[s][0].v));
            vt[s][1] = qf[s][1]; asm volatile("" : "+v"(vt[s][1].v));
        }
    };''')]

_ROT_BEGIN = '''            __syncthreads();                                 // tile `it` visible; the other buffer is free for tile it + 1
            const T* sK = sbase + (it & 1) * fwd_stage_elems<T>();
            const T* sV = sK + KT * LDT;
            const int k0 = it * KT;
            f32x16 s00, s10, s01;
            Frag<T> ka0[4], ka1[4], vt0[2][2], p00[2], p10[2], p11[2];
            load_k(sK, 0, ka0);                              // (ahead of the staging stores in this wave's LDS queue)
            if (it + 1 < ntiles) put(it + 1);
            if (it + 2 < ntiles) fetch(it + 2);'''
VARIANTS["rot_nobarrier"] = [(_ROT_BEGIN, _ROT_BEGIN.replace("            __syncthreads();                                 // tile `it` visible; the other buffer is free for tile it + 1\n", ""))]
VARIANTS["rot_nofetch"] = [(_ROT_BEGIN, _ROT_BEGIN.replace("            if (it + 2 < ntiles) fetch(it + 2);", ""))]
VARIANTS["rot_noput"] = [(_ROT_BEGIN, _ROT_BEGIN.replace("            if (it + 1 < ntiles) put(it + 1);\n", ""))]
VARIANTS["rot_nomem"] = [(_ROT_BEGIN, _ROT_BEGIN.replace("            if (it + 1 < ntiles) put(it + 1);\n", "").replace("            if (it + 2 < ntiles) fetch(it + 2);", "").replace("            __syncthreads();                                 // tile `it` visible; the other buffer is free for tile it + 1\n", ""))]
# score products first in slots B-D (their results are the next slot's first operands)
_SL = [('''            pv(1, p11, vtp);                                 // slot B: PV(prev 1,1) + S(1,0) || softmax(0,0)
            if (TAIL) scores_c(s10, ka0, 1, c0); else scores(s10, ka0, 1);''', '''            if (TAIL) scores_c(s10, ka0, 1, c0); else scores(s10, ka0, 1);
            pv(1, p11, vtp);                                 // slot B: PV(prev 1,1) + S(1,0) || softmax(0,0)'''),
       ('''            pv(0, p00, vt0);                                 // slot C: PV(0,0) + S(0,1) || softmax(1,0)
            if (TAIL) scores_c(s01, ka1, 0, c1); else scores(s01, ka1, 0);''', '''            if (TAIL) scores_c(s01, ka1, 0, c1); else scores(s01, ka1, 0);
            pv(0, p00, vt0);                                 // slot C: PV(0,0) + S(0,1) || softmax(1,0)'''),
       ('''            pv(1, p10, vt0);                                 // slot D: PV(1,0) + S(1,1) || softmax(0,1)
            if (TAIL) scores_c(s11, ka1, 1, c1); else scores(s11, ka1, 1);''', '''            if (TAIL) scores_c(s11, ka1, 1, c1); else scores(s11, ka1, 1);
            pv(1, p10, vt0);                                 // slot D: PV(1,0) + S(1,1) || softmax(0,1)''')]
VARIANTS["rot_sfirst"] = _SL

# l accumulated in four independent chains per query block (the 16 dependent adds of a unit are a latency chain)
_LSPLIT = [('''            for (int t = 0; t < 16; ++t) {
                const float e = fast_exp2(st[t]);
                l[qb] += e;
                st[t] = e;
            }
            pf[0] = frag_from_acc<T>(st, 0);''', '''            for (int t = 0; t < 16; ++t) {
                const float e = fast_exp2(st[t]);
                lp[qb][t & 3] += e;
                st[t] = e;
            }
            pf[0] = frag_from_acc<T>(st, 0);'''),
           ('''        auto soft = [&](f32x16& st, int qb, Frag<T> (&pf)[2]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float e = fast_exp2(st[t]);''', '''        float lp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto soft = [&](f32x16& st, int qb, Frag<T> (&pf)[2]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float e = fast_exp2(st[t]);'''),
           ('''        m[0] = m[1] = 0.f;
    } else {''', '''        m[0] = m[1] = 0.f;
        l[0] = (lp[0][0] + lp[0][1]) + (lp[0][2] + lp[0][3]);
        l[1] = (lp[1][0] + lp[1][1]) + (lp[1][2] + lp[1][3]);
    } else {''')]
VARIANTS["lsplit"] = _LSPLIT
VARIANTS["lsplit_sfirst"] = _LSPLIT + _SL
VARIANTS["lsplit_nomem"] = _LSPLIT + VARIANTS["rot_nomem"]
VARIANTS["nomem_nolds"] = VARIANTS["rot_nomem"] + VARIANTS["fwd_nolds"]
# no key tiles at all: launch + prologue + epilogue of 1024 workgroups
VARIANTS["notiles"] = [("    const int nfull = kvl / KT, ntiles = (kvl + KT - 1) / KT;\n    int ra, rb, col;", "    const int nfull = 0, ntiles = 0;\n    int ra, rb, col;")]
# one workgroup per CU (LDS-forced)
VARIANTS["occ1"] = [("template <typename T> size_t fwd_smem() { return (size_t)2 * fwd_stage_elems<T>() * sizeof(T); }",
                     "template <typename T> size_t fwd_smem() { return (size_t)90 * 1024; }")]

"""Experiment of the day for tools/attn_lab.py (timing only; several of these compute wrong results by design)."""
VARIANTS = {}
# ---- dK/dV kernel ablations
_DKDV_LOOP = '''            __syncthreads();                   // tile `it` visible; the other stage is free for tile it + 1
            if (it + 1 < nq) put(it + 1);
            if (it + 2 < nq) fetch(it + 2);
            if (kw0 < kvl) {                   // wave-uniform'''
VARIANTS["kv_nobarrier"] = [(_DKDV_LOOP, _DKDV_LOOP.replace("            __syncthreads();                   // tile `it` visible; the other stage is free for tile it + 1\n", ""))]
VARIANTS["kv_nofetch"] = [(_DKDV_LOOP, _DKDV_LOOP.replace("            if (it + 2 < nq) fetch(it + 2);\n", ""))]
VARIANTS["kv_noput"] = [(_DKDV_LOOP, _DKDV_LOOP.replace("            if (it + 1 < nq) put(it + 1);\n", ""))]
VARIANTS["kv_noexp"] = [('''                        const float pv = fast_exp2(st[t]);
                        st[t] = pv;
                        ds[t] *= pv;''', '''                        const float pv = st[t] * 0.5f;
                        st[t] = pv;
                        ds[t] *= pv;''')]
VARIANTS["kv_norowconst"] = [('''                    load_rowconst(0, cL0, cD0);
                    load_rows(0, qa0, oa0);
                    load_rowconst(1, cL1, cD1);''', '''                    cL0 = f32x16{0}; cD0 = f32x16{0}; cL1 = f32x16{0}; cD1 = f32x16{0};
                    asm volatile("" : "+v"(cL0), "+v"(cD0), "+v"(cL1), "+v"(cD1));
                    load_rows(0, qa0, oa0);''')]
VARIANTS["kv_occ1"] = [("template <typename T> size_t dkdv_smem() { return (size_t)2 * dkdv_stage_bytes<T>(); }",
                        "template <typename T> size_t dkdv_smem() { return (size_t)90 * 1024; }")]
VARIANTS["kv_occ3"] = [("__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_bwd_dkdv_kernel(",
                        "__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 3 : 1)) void attn_bwd_dkdv_kernel(")]
VARIANTS["kv_nosgb"] = [('''                        for (int g = 0; g < 8; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }''', '''                        for (int g = 0; g < 0; ++g) {
                        }''')]

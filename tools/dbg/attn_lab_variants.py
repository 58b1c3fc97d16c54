"""Experiment of the day for tools/attn_lab.py (timing only; several of these compute wrong results by design)."""
TB = '''    auto tile_begin = [&](int it) {
        __syncthreads();
        if (it + 1 < ntiles) put(it + 1);
        if (it + 2 < ntiles) fetch(it + 2);
    };'''
VARIANTS = {
    "nobarrier": [(TB, TB.replace("        __syncthreads();\n", ""))],
    "nofetch": [(TB, TB.replace("        if (it + 2 < ntiles) fetch(it + 2);\n", ""))],
    "noput": [(TB, TB.replace("        if (it + 1 < ntiles) put(it + 1);\n", ""))],
    "noexp": [("                const float e = fast_exp2(st[t]);\n                l[qb] += e;", "                const float e = st[t] * 0.5f;\n                l[qb] += e;")],
    "nosgb": [('''                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }''', '''                for (int g = 0; g < 0; ++g) {
                }''')],
}

# per-workgroup start / end (100 MHz s_memrealtime) + hardware id, read back by attn_lab.py `timeline`
_TL_DECL = '''namespace {

__device__ unsigned long long g_tl[4 * 8192];
__device__ unsigned long long g_tl2[4 * 8192];
'''
_TL_END = '''    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: in-order LDS, no barrier needed
    constexpr int CH = DH * (int)sizeof(T) / 16, RPP = 64 / CH;       // 16-byte chunks per row, rows per pass'''
VARIANTS["timeline"] = [
    ("namespace {\n", _TL_DECL),
    ('''    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    // all keys masked -> the reference's masked_fill(-65504) + softmax gives the uniform average over all N
    // keys: run the ordinary loop with Q = 0 (every score 0, every p = 1).''',
     '''    const int r = lane & 31, half = lane >> 5;
    const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;'''),
    (_TL_END, '''    if (tid == 0 && blockIdx.x < 8192) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_tl[4 * blockIdx.x] = tl0; g_tl[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        g_tl[4 * blockIdx.x + 2] = hw; g_tl[4 * blockIdx.x + 3] = xcc;
        g_tl2[4 * blockIdx.x] = tl1; g_tl2[4 * blockIdx.x + 1] = tl2;
    }
''' + _TL_END),
    ('''    fetch(0);
    put(0);
    if (ntiles > 1) fetch(1);''', '''    fetch(0);
    put(0);
    if (ntiles > 1) fetch(1);
    const unsigned long long tl1 = __builtin_amdgcn_s_memrealtime();'''),
    ('''    // Epilogue.  A lane owns one query and 4-element pieces of its O row, so direct stores would be 8-byte pieces''',
     '''    const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
    // Epilogue.  A lane owns one query and 4-element pieces of its O row, so direct stores would be 8-byte pieces'''),
    ('''    if (qrow < p.N && half == 0) p.lse[((size_t)b * p.H + hd) * p.N + qrow] = m[qb] + log2f(l[qb]);
    }
}''' if False else '''                *reinterpret_cast<u32x4_t*>(p.o_res + off) = sv;
            }
        }
    }
}''', '''                *reinterpret_cast<u32x4_t*>(p.o_res + off) = sv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x < 8192) g_tl2[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
}'''),
    ('''extern "C" long long mtmp_key_norms_floats(long long rows, int H)''',
     '''extern "C" int mtmp_debug_timeline(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), n * sizeof(unsigned long long)) != hipSuccess;
}
extern "C" int mtmp_debug_timeline2(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl2), n * sizeof(unsigned long long)) != hipSuccess;
}
extern "C" long long mtmp_key_norms_floats(long long rows, int H)'''),
]

# one workgroup per CU (LDS request > half of 160 KiB): what does a wave do when it has its SIMD to itself?
VARIANTS["occ1"] = [("template <typename T> size_t fwd_smem() { return (size_t)2 * fwd_stage_elems<T>() * sizeof(T); }",
                     "template <typename T> size_t fwd_smem() { return (size_t)90 * 1024; }")]

# s_memtime stamps around the phases of the bounded body (sums per wave, averaged by attn_lab.py `stamps`)
_ST_DECL = '''namespace {

__device__ unsigned long long g_st[16];
#define STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); }
'''
_BODY_OLD = '''        auto body = [&](int it, auto tail_tag) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            tile_begin(it);
            const T* sK = sbase + (it & 1) * fwd_stage_elems<T>();
            const T* sV = sK + KT * LDT;
            const int k0 = it * KT;
            f32x16 s00, s10, s01, s11;                       // s<qb><kb>'''
_BODY_NEW = '''        unsigned long long sa[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto body = [&](int it, auto tail_tag) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            unsigned long long t0, t1, t2, t3, t4, t5, t6, t7, t8;
            STAMP(t0)
            __syncthreads();
            STAMP(t1)
            if (it + 1 < ntiles) put(it + 1);
            unsigned long long t1b;
            STAMP(t1b)
            sa[9] += t1b - t1;
            if (it + 2 < ntiles) fetch(it + 2);
            STAMP(t2)
            const T* sK = sbase + (it & 1) * fwd_stage_elems<T>();
            const T* sV = sK + KT * LDT;
            const int k0 = it * KT;
            f32x16 s00, s10, s01, s11;                       // s<qb><kb>'''
def _stamp_after(marker, name, first=False):
    return (marker, marker.replace("__builtin_amdgcn_sched_barrier(0);", f"STAMP({name})", 1))
VARIANTS["stamps"] = [
    ("namespace {\n", _ST_DECL),
    (_BODY_OLD, _BODY_NEW),
    ('''            if (TAIL) mask_tail(s00, k0);
            __builtin_amdgcn_sched_barrier(0);''', '''            if (TAIL) mask_tail(s00, k0);
            STAMP(t3)'''),
    ('''            load_v(sV, 0, vt0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);''', '''            load_v(sV, 0, vt0);
            interleave();
            STAMP(t4)'''),
    ('''            soft(s10, 1, p10);
            interleave();
            __builtin_amdgcn_sched_barrier(0);''', '''            soft(s10, 1, p10);
            interleave();
            STAMP(t5)'''),
    ('''            load_v(sV, 1, vt1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);''', '''            load_v(sV, 1, vt1);
            interleave();
            STAMP(t6)'''),
    ('''            soft(s11, 1, p11);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            pv(1, p11, vt1);                                 // slot 5
        };
        for (int it = 0; it < nfull; ++it) body(it, std::false_type{});''', '''            soft(s11, 1, p11);
            interleave();
            STAMP(t7)
            pv(1, p11, vt1);                                 // slot 5
            STAMP(t8)
            sa[0] += t1 - t0; sa[1] += t2 - t1; sa[2] += t3 - t2; sa[3] += t4 - t3; sa[4] += t5 - t4; sa[5] += t6 - t5;
            sa[6] += t7 - t6; sa[7] += t8 - t7; sa[8] += 1;
        };
        for (int it = 0; it < nfull; ++it) body(it, std::false_type{});'''),
    ('''        if (ntiles > nfull) body(nfull, std::true_type{});
        m[0] = m[1] = 0.f;''', '''        if (ntiles > nfull) body(nfull, std::true_type{});
        if (lane == 0) for (int i = 0; i < 10; ++i) atomicAdd(&g_st[i], sa[i]);
        m[0] = m[1] = 0.f;'''),
    ('''extern "C" long long mtmp_key_norms_floats(long long rows, int H)''',
     '''extern "C" int mtmp_debug_stamps(unsigned long long* out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_st), z, sizeof(z)) != hipSuccess;
}
extern "C" long long mtmp_key_norms_floats(long long rows, int H)'''),
    ('''    auto tile_begin = [&](int it) {
        __syncthreads();
        if (it + 1 < ntiles) put(it + 1);
        if (it + 2 < ntiles) fetch(it + 2);
    };''', '''    auto tile_begin = [&](int it) {
        __syncthreads();
        if (it + 1 < ntiles) put(it + 1);
        if (it + 2 < ntiles) fetch(it + 2);
    };
    (void)tile_begin;'''),
]
VARIANTS["stamps_occ1"] = VARIANTS["stamps"] + VARIANTS["occ1"]

VARIANTS["stamps_nofetch"] = [(a, b.replace("            if (it + 2 < ntiles) fetch(it + 2);\n", "")) for a, b in VARIANTS["stamps"]]

# ---- instruction-order experiments on the bounded body
_SGB = '''                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }'''
# exponentials in the first half of the slot, adds + conversions in the second (nothing reads a fresh exp)
VARIANTS["trans_first"] = [(_SGB, '''                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                }
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }''')]
# start stagger: the workgroup in the second wave slot of its SIMD sleeps ~1 us before its first tile
_STG_OLD = '''    fetch(0);
    put(0);
    if (ntiles > 1) fetch(1);'''
VARIANTS["stagger"] = [(_STG_OLD, '''    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (hw & 1) { for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127); }
    }
''' + _STG_OLD)]
VARIANTS["prio_slot"] = [(_STG_OLD, '''    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (hw & 1) __builtin_amdgcn_s_setprio(1);
    }
''' + _STG_OLD)]
# priority alternating per tile, opposite in the two wave slots of a SIMD
VARIANTS["prio_alt"] = [(_STG_OLD, '''    unsigned hwslot;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwslot));
    hwslot &= 1;
''' + _STG_OLD), ('''    auto tile_begin = [&](int it) {
        __syncthreads();''', '''    auto tile_begin = [&](int it) {
        if ((it + hwslot) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        __syncthreads();''')]

# scratch: VARIANTS = {name: [(file, old, new), ...]} for tools/lab_lib.py (patched copies of csrc/, timing experiments only)
# round 5's experiments (GELU of swin_mlp as a packed polynomial, gemm_nt fragment loads pipelined at 3 workgroups per CU) are in
# DESIGN.md section 9; what is kept here is the build with the in-kernel clock stamps (tools/dbg/swin_mlp_clock.py, dkdv64_clock.py).
_PK = [
    ("attention.hip", "    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};",
     "    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};\n    typedef float f32x2_t __attribute__((ext_vector_type(2)));\n    f32x2_t lp[2] = {{0.f, 0.f}, {0.f, 0.f}};"),
    ("attention.hip", """            for (int t = 0; t < 16; ++t) {
                const float e = fast_exp2(st[t]);
                l[qb] += e;
                st[t] = e;
            }
            pf[0] = frag_from_acc<T>(st, 0);""", """            for (int t = 0; t < 16; t += 2) {
                const float e0 = fast_exp2(st[t]), e1 = fast_exp2(st[t + 1]);
                lp[qb] += (f32x2_t){e0, e1};
                st[t] = e0;
                st[t + 1] = e1;
            }
            pf[0] = frag_from_acc<T>(st, 0);"""),
    ("attention.hip", """        l[qb] = half_sum(l[qb]);
        const float inv = 1.0f / l[qb];""", """        l[qb] = half_sum(l[qb] + (lp[qb][0] + lp[qb][1]));
        const float inv = 1.0f / l[qb];"""),
]
VARIANTS = {
    "clock": [("Makefile", "-Wno-unused-result -mllvm", "-Wno-unused-result -DMTMP_LAB_CLOCK -mllvm")],
    "fwd_pk": list(_PK),
    "fwd_pk4": list(_PK) + [("attention.hip", """                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            }
        };
        // The unit pipeline runs ACROSS""", """                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
        };
        // The unit pipeline runs ACROSS""")],
}


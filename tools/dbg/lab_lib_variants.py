VARIANTS = {
    "lnb_noepi": [("gemm.hip", "    for (int trip = 0; trip < 32 / RPW; ++trip) {\n        const int rl0 = wave * 32 + trip * RPW;\n        if (m0 + rl0 >= M) break;                        // wave-uniform: whole trips past M do nothing",
                   "    for (int trip = 0; trip < 0; ++trip) {\n        const int rl0 = wave * 32 + trip * RPW;\n        if (m0 + rl0 >= M) break;")],
    "lnb_nomma": [("gemm.hip", "                for (int rt = 0; rt < 2; ++rt) mma<T>(acc[rt][nt], w, a[rt]);\n            }\n        }\n    }\n    __syncthreads();                          // every wave is done with sA / sW: the dxn tile takes their place",
                   "                for (int rt = 0; rt < 2; ++rt) acc[rt][nt][0] += to_f32(w.v[0]) * to_f32(a[rt].v[0]);\n            }\n        }\n    }\n    __syncthreads();")],
    "lnb_nokloop": [("gemm.hip", "    for (int kc = 0; kc < nk; ++kc) {\n        __syncthreads();\n        tile_commit<T>(sA, areg, tid);\n        tile_commit<T>(sW, wreg0, tid);\n        tile_commit<T>(sW + 128 * LDW, wreg1, tid);",
                     "    for (int kc = 0; kc < 1; ++kc) {\n        __syncthreads();\n        tile_commit<T>(sA, areg, tid);\n        tile_commit<T>(sW, wreg0, tid);\n        tile_commit<T>(sW + 128 * LDW, wreg1, tid);")],
}

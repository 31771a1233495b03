import sys
p='/tmp/sw/b/music_amd/csrc/wn_epilogue.hip'
s=open('/root/repo/music_amd/csrc/wn_epilogue.hip').read()
def rep(old,new,cnt=1):
    global s
    assert old in s, old[:60]
    s=s.replace(old,new,cnt)
rep('#define EPI_COLS 128\n','''#define EPI_COLS 128
// timing builds (-DEPI_T=n; wrong results): 1 no MFMAs, 2 no weight requests in the skip loop, 3 no z requests after the
// prologue, 4 no barrier in the skip loop, 6 no stores, 7 no post-processing products, 8 no split / LDS fill in the skip loop
#ifndef EPI_T
#define EPI_T 0
#endif
#ifdef EPI_DBG
// phase clock sums (developer build, tools/epi_clocks.py): per wave half [0-3 | 4-7]: weight requests, first k-step, split + fill + z
// requests, second k-step, barrier; then the whole loop, the post-processing part, the kernel
__device__ unsigned long long epi_dbg[32];
#define EPI_TICK(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
extern "C" int wn_epi_dbg_read(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(epi_dbg), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(epi_dbg), z, sizeof(z));
    }
    return (int)e;
}
#endif
''')
rep("            for (int i = 0; i < 2; ++i) mma<T, NS>(acc[i][n], af[i], bf[n % 3]);\n","""            for (int i = 0; i < 2; ++i) {
                if (EPI_T == 1) asm volatile("" :: "v"(af[i].hi), "v"(af[i].lo), "v"(bf[n % 3].hi), "v"(bf[n % 3].lo));
                else mma<T, NS>(acc[i][n], af[i], bf[n % 3]);
            }
""")
rep("""        const int itw = it + 2 < NI ? it + 2 : NI - 1, itr = it + 3 < NI ? it + 3 : NI - 1;
""","""        const int itw = it + 2 < NI ? it + 2 : NI - 1, itr = it + 3 < NI ? it + 3 : NI - 1;
#ifdef EPI_DBG
        EPI_TICK(c0);
        const unsigned long long c1 = c0;
#endif
""")
rep("""            if (n & 1) afn[kk][i].lo = __builtin_bit_cast(typename T::vec8, *p);
            else afn[kk][i].hi = __builtin_bit_cast(typename T::vec8, *p);
        });
        __builtin_amdgcn_sched_barrier(0);
        store_raw(rnext, (it + 1) & 1);                       // (the last iteration fills a stage nobody reads)
        __builtin_amdgcn_sched_barrier(0);
""","""            if (EPI_T == 2) return;
            if (n & 1) afn[kk][i].lo = __builtin_bit_cast(typename T::vec8, *p);
            else afn[kk][i].hi = __builtin_bit_cast(typename T::vec8, *p);
        });
        __builtin_amdgcn_sched_barrier(0);
#ifdef EPI_DBG
        EPI_TICK(c2);
#endif
        if (EPI_T != 8) store_raw(rnext, (it + 1) & 1);                       // (the last iteration fills a stage nobody reads)
        else asm volatile("" :: "v"(rnext[0]), "v"(rnext[1]), "v"(rnext[2]), "v"(rnext[3]));
        __builtin_amdgcn_sched_barrier(0);
#ifdef EPI_DBG
        EPI_TICK(c3);
#endif
""")
rep("""            if (n < 4) rnext[n] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zrow + (size_t)itr * 64 * a.pitch + (size_t)n * a.pitch));
        });
        __syncthreads();""","""            if (EPI_T != 3 && n < 4) rnext[n] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zrow + (size_t)itr * 64 * a.pitch + (size_t)n * a.pitch));
        });
#ifdef EPI_DBG
        EPI_TICK(c4);
#endif
        if (EPI_T != 4) __syncthreads();
#ifdef EPI_DBG
        EPI_TICK(c5);
        dbg_acc[0] += c1 - c0; dbg_acc[1] += c2 - c1; dbg_acc[2] += c3 - c2; dbg_acc[3] += c4 - c3; dbg_acc[4] += c5 - c4;
#endif""")
rep("        load_w2(af1, a.w_skip, KS, it0 + 1);\n","        load_w2(af1, a.w_skip, KS, it0 + 1);\n        if (EPI_T == 2) load_w2(af2, a.w_skip, KS, it0 + 2);\n")
rep("    init_acc(acc, a.bias_s, a.s_valid);\n","#ifdef EPI_DBG\n    unsigned long long dbg_acc[5] = {};\n    const unsigned long long k0_ = __builtin_readcyclecounter();\n#endif\n    init_acc(acc, a.bias_s, a.s_valid);\n")
rep("    if (it0 < NI) {\n","#ifdef EPI_DBG\n    const unsigned long long l0_ = __builtin_readcyclecounter();\n#endif\n    if (it0 < NI) {\n")
rep("    const bool tile_in = tile0 >= a.t_lo && tile0 + EPI_COLS <= a.t_hi;\n","#ifdef EPI_DBG\n    const unsigned long long l1_ = __builtin_readcyclecounter();\n#endif\n    const bool tile_in = tile0 >= a.t_lo && tile0 + EPI_COLS <= a.t_hi;\n")
old="        float* out = base + (size_t)b * bstride + (size_t)(m0 * 16 + 4 * q) * pitch + tile0 + 4 * c + shift;\n"
rep(old,old+"        if (EPI_T == 6) {\n#pragma unroll\n            for (int i = 0; i < 2; ++i)\n#pragma unroll\n                for (int n = 0; n < 8; ++n) asm volatile(\"\" :: \"v\"(acc[i][n]));\n            return;\n        }\n")
old="#pragma unroll\n    for (int s = 0; s < 2; ++s) request(wf[s], a.w_p1c, s);\n"
rep(old,"    if (EPI_T == 7) { store_rows(acc, 0, 16, a.u, a.s_bstride, a.pitch, 0, a.s_valid); return; }\n"+old)
old="    store_rows(acc, 0, 16, a.o, a.o_bstride, a.o_pitch, -a.t_lo, a.q_valid);\n}"
rep(old,old[:-1]+'''#ifdef EPI_DBG
    const unsigned long long k1_ = __builtin_readcyclecounter();
    if (lane == 0) {
        for (int z_ = 0; z_ < 5; ++z_) atomicAdd(&epi_dbg[(wave >> 2) * 8 + z_], dbg_acc[z_]);
        atomicAdd(&epi_dbg[(wave >> 2) * 8 + 5], l1_ - l0_);
        atomicAdd(&epi_dbg[(wave >> 2) * 8 + 6], k1_ - l1_);
        atomicAdd(&epi_dbg[(wave >> 2) * 8 + 7], k1_ - k0_);
        if (wave == 0) atomicAdd(&epi_dbg[16], 1ull);
    }
#endif
}''')
open(p,'w').write(s)
print("instrumented")

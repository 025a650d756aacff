// store_hazard.hip — the VMEM store write-data hazard behind the round-2 exchange rewrite's rare 1e-7 errors, in isolation.
//
//   buffer_store_dwordx4 v[D:D+3], voff, rsrc, sOFF offen      ; 16-byte store, soffset in an SGPR
//   v_mov_b32            vD, <something else>                   ; VALU overwrites the first data register in the next issue slot
//
// The ISA manuals list "VMEM store of more than 64 bits followed by a VALU write of the VGPRs holding the write data: 1 wait state"
// with the exemption "BUFFER_STORE_* operations that use an SGPR for the offset do not require any wait states", and the compiler
// follows it (LLVM GCNHazardRecognizer::createsVALUHazard: no hazard when soffset is a register) — with an immediate soffset hipcc
// inserts s_nop 1, with an SGPR soffset it does not.  This probe issues exactly that pair from every lane, NOPS wait states apart,
// and counts the 16-byte records whose first dword is the overwriting value instead of the stored one.  Run it alone and with
// several copies at once (the errors need pressure on the memory pipeline).
//   hipcc --offload-arch=gfx950 -O3 tools/store_hazard.hip -o build/store_hazard && build/store_hazard [rounds]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

using rsrc_t = __amdgpu_buffer_rsrc_t;

template <int NOPS, bool SGPR_SOFF>
__global__ void __launch_bounds__(256) k_probe(uint32_t* out, uint32_t n_records, uint32_t rounds)
{
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, n_records * 16u, 0x00020000);
    const uint32_t tid = threadIdx.x;
    const uint32_t per_block = 256u * rounds;
    uint32_t base = blockIdx.x * per_block;  // uniform
    for (uint32_t k = 0; k < rounds; ++k, base += 256u) {
        const uint32_t rec = base + tid;
        const uint32_t a0 = 0xA0000000u | rec, a1 = 0xB0000000u | rec, a2 = 0xC0000000u | rec, a3 = 0xD0000000u | rec;
        const uint32_t voff = tid * 16u;
        const uint32_t soff = base * 16u;
        const uint32_t vo = SGPR_SOFF ? voff : rec * 16u;
        const uint32_t clobber = 0xDEAD0000u | (rec & 0xFFFFu);
        // the store and, NOPS wait states later, a VALU write of its first data register — one asm block with hard registers, so that
        // no compiler pass can separate or pad the pair
#define PROBE_ASM(STORE, GAP)                                                                                               \
    asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\ts_nop 4\n\t" STORE "\n\t" GAP \
                 "v_mov_b32 v100, %7\n\ts_nop 4"                                                                             \
                 :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(vo), "s"(r), "s"(soff), "v"(clobber) : "memory", "v100", "v101", "v102", "v103")
        if (SGPR_SOFF) {
            if (NOPS == 0) PROBE_ASM("buffer_store_dwordx4 v[100:103], %4, %5, %6 offen", "");
            else if (NOPS == 1) PROBE_ASM("buffer_store_dwordx4 v[100:103], %4, %5, %6 offen", "s_nop 0\n\t");
            else PROBE_ASM("buffer_store_dwordx4 v[100:103], %4, %5, %6 offen", "s_nop 1\n\t");
        } else {
            if (NOPS == 0) PROBE_ASM("buffer_store_dwordx4 v[100:103], %4, %5, 0 offen", "");
            else if (NOPS == 1) PROBE_ASM("buffer_store_dwordx4 v[100:103], %4, %5, 0 offen", "s_nop 0\n\t");
            else PROBE_ASM("buffer_store_dwordx4 v[100:103], %4, %5, 0 offen", "s_nop 1\n\t");
        }
#undef PROBE_ASM
    }
}

template <int NOPS, bool SGPR_SOFF>
static void run(const char* name, uint32_t* d, std::vector<uint32_t>& h, uint32_t blocks, uint32_t rounds, int reps)
{
    const uint32_t n = blocks * 256u * rounds;
    uint64_t bad_first = 0, bad_other = 0, total = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(d, 0, (size_t)n * 16);
        hipLaunchKernelGGL((k_probe<NOPS, SGPR_SOFF>), dim3(blocks), dim3(256), 0, 0, d, n, rounds);
        hipMemcpy(h.data(), d, (size_t)n * 16, hipMemcpyDeviceToHost);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t* p = &h[(size_t)i * 4];
            if (p[0] != (0xA0000000u | i)) ++bad_first;
            if (p[1] != (0xB0000000u | i) || p[2] != (0xC0000000u | i) || p[3] != (0xD0000000u | i)) ++bad_other;
        }
        total += n;
    }
    std::printf("%-44s records %llu  first dword wrong %llu  other dwords wrong %llu\n", name, (unsigned long long)total,
                (unsigned long long)bad_first, (unsigned long long)bad_other);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
    const uint32_t blocks = 2048, rounds = 64;
    const uint32_t n = blocks * 256u * rounds;
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (size_t)n * 16) != hipSuccess) { std::puts("no device"); return 1; }
    std::vector<uint32_t> h((size_t)n * 4);
    run<0, true>("SGPR soffset, 0 wait states", d, h, blocks, rounds, reps);
    run<1, true>("SGPR soffset, s_nop 0 (1 wait state)", d, h, blocks, rounds, reps);
    run<2, true>("SGPR soffset, s_nop 1 (2 wait states)", d, h, blocks, rounds, reps);
    run<0, false>("immediate soffset 0, 0 wait states", d, h, blocks, rounds, reps);
    run<2, false>("immediate soffset 0, s_nop 1", d, h, blocks, rounds, reps);
    hipFree(d);
    return 0;
}

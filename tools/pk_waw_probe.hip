// Probe: packed fp32 adds on a SIMD shared with waves that keep the matrix pipe busy (gfx950).
// hipcc's SLP vectoriser turned the crop gather's (l*c - w*s) + x, (l*s + w*c) + y into v_pk_add_f32 with
// op_sel / neg modifiers; with two conv1 workgroups per CU lanes 48-63 of some gathers fetched the wrong pixel.
// This replays such instructions from registers and checks every lane against the scalar result.
// Build: hipcc --offload-arch=gfx950 -O2 tools/pk_waw_probe.hip -o tools/pk_waw_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LOADREGS "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %7\n\ts_nop 7\n\t"
#define STOREREGS(a, b) "s_nop 7\n\tv_mov_b32 %0, " a "\n\tv_mov_b32 %1, " b "\n\t"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

template <int VARIANT, bool MFMA_NEIGHBOURS>
__global__ __launch_bounds__(512, 4) void probe(int iters, unsigned* bad_lanes /*64 counters*/, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave & 1) {
        if (!MFMA_NEIGHBOURS) return;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane - j); }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters * 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        if (c[0] == 12345.f) sink[0] = c[1];
        return;
    }
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        const float lc = (float)(lane + i) * 0.37f, ls = (float)(lane * 3 + i) * 0.11f;
        const float wc = (float)(lane ^ 21) * 0.23f, ws = (float)(lane + 7) * 0.41f, x = 100.5f, y = -37.25f;
        float r0, r1, e0, e1;
        if (VARIANT == 0) {          // the compiled sequence
            asm volatile(LOADREGS
                "v_pk_add_f32 v[24:25], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "v_pk_add_f32 v[20:21], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "s_nop 0\n\tv_mov_b32 v21, v25\n\t"
                "v_pk_add_f32 v[20:21], v[26:27], v[20:21]\n\t" STOREREGS("v20", "v21")
                : "=v"(r0), "=v"(r1) : "v"(lc), "v"(ls), "v"(wc), "v"(ws), "v"(x), "v"(y) : CLOB);
            e0 = __fadd_rn(__fsub_rn(lc, ws), x); e1 = __fadd_rn(__fadd_rn(ls, wc), y);
        } else if (VARIANT == 1) {   // one plain packed add
            asm volatile(LOADREGS "v_pk_add_f32 v[24:25], v[20:21], v[22:23]\n\t" STOREREGS("v24", "v25")
                : "=v"(r0), "=v"(r1) : "v"(lc), "v"(ls), "v"(wc), "v"(ws), "v"(x), "v"(y) : CLOB);
            e0 = __fadd_rn(lc, wc); e1 = __fadd_rn(ls, ws);
        } else if (VARIANT == 2) {   // one packed add with crossed halves
            asm volatile(LOADREGS "v_pk_add_f32 v[24:25], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]\n\t" STOREREGS("v24", "v25")
                : "=v"(r0), "=v"(r1) : "v"(lc), "v"(ls), "v"(wc), "v"(ws), "v"(x), "v"(y) : CLOB);
            e0 = __fadd_rn(lc, ws); e1 = __fadd_rn(ls, wc);
        } else if (VARIANT == 3) {   // crossed halves + negated second operand
            asm volatile(LOADREGS "v_pk_add_f32 v[24:25], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" STOREREGS("v24", "v25")
                : "=v"(r0), "=v"(r1) : "v"(lc), "v"(ls), "v"(wc), "v"(ws), "v"(x), "v"(y) : CLOB);
            e0 = __fsub_rn(lc, ws); e1 = __fsub_rn(ls, wc);
        } else if (VARIANT == 4) {   // control: two scalar adds
            asm volatile(LOADREGS "v_add_f32 v24, v20, v22\n\tv_add_f32 v25, v21, v23\n\t" STOREREGS("v24", "v25")
                : "=v"(r0), "=v"(r1) : "v"(lc), "v"(ls), "v"(wc), "v"(ws), "v"(x), "v"(y) : CLOB);
            e0 = __fadd_rn(lc, wc); e1 = __fadd_rn(ls, ws);
        } else {                     // packed multiply
            asm volatile(LOADREGS "v_pk_mul_f32 v[24:25], v[20:21], v[22:23]\n\t" STOREREGS("v24", "v25")
                : "=v"(r0), "=v"(r1) : "v"(lc), "v"(ls), "v"(wc), "v"(ws), "v"(x), "v"(y) : CLOB);
            e0 = __fmul_rn(lc, wc); e1 = __fmul_rn(ls, ws);
        }
        if (r0 != e0 || r1 != e1) nbad++;
    }
    if (nbad) atomicAdd(&bad_lanes[lane], nbad);
}

template <int V, bool M>
static void run(const char* name, unsigned* bad, float* sink) {
    for (int grid : {256, 512, 1024}) {
        (void)hipMemset(bad, 0, 256);
        probe<V, M><<<grid, 512>>>(20000, bad, sink);
        hipError_t e = hipDeviceSynchronize();
        unsigned h[64];
        (void)hipMemcpy(h, bad, 256, hipMemcpyDeviceToHost);
        unsigned long long q[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
        printf("%-44s grid %4d (%d WG/CU): err %d, wrong per lane quad: %llu %llu %llu %llu\n", name, grid, grid / 256, (int)e, q[0], q[1], q[2], q[3]);
    }
}

int main() {
    unsigned* bad; float* sink;
    (void)hipMalloc(&bad, 256); (void)hipMalloc(&sink, 4);
    run<0, true>("compiled sequence, MFMA neighbours", bad, sink);
    run<0, false>("compiled sequence, idle neighbours", bad, sink);
    run<1, true>("v_pk_add_f32 plain, MFMA neighbours", bad, sink);
    run<2, true>("v_pk_add_f32 op_sel crossed, MFMA neighbours", bad, sink);
    run<3, true>("v_pk_add_f32 crossed+neg, MFMA neighbours", bad, sink);
    run<3, false>("v_pk_add_f32 crossed+neg, idle neighbours", bad, sink);
    run<4, true>("2 x v_add_f32 (control), MFMA neighbours", bad, sink);
    run<5, true>("v_pk_mul_f32 plain, MFMA neighbours", bad, sink);
    return 0;
}

// conv_h2b.hip -- a whole 32-channel BasicBlock (simple_romp/romp/model.py:54-83) in ONE kernel on the f16x2 split:
//     y = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x),   32 -> 32 -> 32 channels, stride 1, H2 tensors in and out.
// Why (round 3, profiles/r03_h2r_notes.md): the 32-channel @128^2 class is the network's largest (64 launches per image batch,
// ~3.1 ms of a 14.9 ms forward) and its time does not respond to anything done INSIDE a conv kernel; as two launches a block
// moves 5 tensors over the fabric (x -> m; m, x -> y: 335 MB at B = 32) and pays launch / prologue / epilogue twice.  Fused, x is
// read once (haloed), y written once, and the intermediate m never leaves the CU.
//
// One 256-thread workgroup per CU -- ONE wave per SIMD, so each wave owns the SIMD's whole 512-entry register file -- persistent
// over 16x16-pixel output tiles.  Every wave keeps the split weights of BOTH convs in registers for the whole launch (2 x 144
// VGPRs: no weight traffic after the prologue).  Per tile:
//   1. conv1 on the tile's 18x18 halo of m (11 blocks of 32 pixels: nine 2-row x 16-column blocks and two blocks holding the
//      edge columns; 3 block slots per wave) from the 20x20 input halo of x, which arrived by LDS-DMA (both 16-channel chunks)
//      while the previous tile was in steps 3-4; the residual rows of this tile are loaded into registers first;
//   2. barrier; the DMA of the NEXT tile's halo is issued (it has steps 3 and 4 to land); bn1 + ReLU, ZERO outside the image
//      (conv2's padding), pre-split, written into LDS in the rotated unit layout the fragment reads use; barrier;
//   3. conv2 from m in LDS (two blocks per wave);
//   4. the ordinary fused epilogue (bn2 + x + ReLU, H2 stores); drain; barrier.
// With one wave per SIMD nothing hides a wave's own stalls, so the fragment reads run three block-steps ahead of their MFMAs
// (registers are not scarce here) and the only waits on memory sit at the two ends of a tile.
#include "conv_split.h"
#include <string.h>

namespace romp {

struct BCfg {
    static constexpr int TH = 16, TW = 16;
    static constexpr int IR = TH + 4, IC = TW + 4;             // input halo 20 x 20
    static constexpr int MR = TH + 2, MC = TW + 2;             // intermediate halo 18 x 18
    static constexpr int RSU = 80;                             // 16-byte units per (input or m) row: 20 pixel slots x 4 units
    static constexpr int NI = 7;                               // DMA pieces per wave and 16-channel stage: 4 x 7 x 64 units >= 20 x 80
    static constexpr int STAGE_BYTES = 4 * NI * 1024;          // 28 KiB (1600 units used), one per input chunk
    static constexpr int MID_PLANE = MR * RSU * 16;            // one 16-channel chunk of m: 23 040 bytes
    static constexpr int OFF_M = 2 * STAGE_BYTES;
    static constexpr int OFF_E = OFF_M + 2 * MID_PLANE;        // epilogue staging tiles, one per wave
    static constexpr int OFF_S = OFF_E + 4 * EPI_WAVE;         // [scale1 32 | shift1 32 | scale2 32 | shift2 32]
    static constexpr int LDS_BYTES = OFF_S + 128 * 4 + 16;
    static_assert(IR * RSU <= 4 * NI * 64, "stage pieces cover the input halo");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

typedef __attribute__((address_space(3))) void lds_void_b;
typedef const __attribute__((address_space(1))) void glb_void_b;

// unit index of (column c, unit w of the 16-channel chunk) inside a row of RSU units (the rotated layout of conv_h2r.hip)
__device__ __forceinline__ int unit_of(int c, int w) { return (c >> 2) * 16 + (c & 3) + 4 * ((w + (c >> 2)) & 3); }

// ConvParams as used here: in = x (H2), res = x, out = y (H2); w3 = conv1's split weights, wh = conv2's; scale = conv1's
// f16x2 epilogue scale (32), w = conv1's shift (32, as floats), scale_h / shift = conv2's; the geometry fields as for a conv.
__global__ __launch_bounds__(256, 1) void bblock32_kernel(ConvParams p) {
    using X = BCfg;
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    char* sM = sBuf + X::OFF_M;
    float* sS = reinterpret_cast<float*>(sBuf + X::OFF_S);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_b*)sBuf;      // LDS byte address of the dynamic segment

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int nwg_q = gridDim.x / p.n_queues;
    const int j0 = blockIdx.x / p.n_queues;
    if (j0 >= p.per_queue) return;
    const int n_mine = (p.per_queue - j0 + nwg_q - 1) / nwg_q;  // tiles of this workgroup: j0, j0 + nwg_q, ...

    if (tid < 128) {
        const float* src = tid < 32 ? p.scale : tid < 64 ? p.w : tid < 96 ? p.scale_h : p.shift;
        sS[tid] = src[tid & 31];
    }
    // ---- the weights of both convs, all taps and both chunks, resident: lane (li, lh) = channel li, k-half lh
    frag w1[2][9][2], w2[2][9][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {                    // packed [tap][chunk 2][piece 2][k-half 2][32] units
                w1[ch][tap][pc] = __builtin_bit_cast(frag, p.w3[(((tap * 2 + ch) * 2 + pc) * 2 + lh) * 32 + li]);
                w2[ch][tap][pc] = __builtin_bit_cast(frag, p.wh[(((tap * 2 + ch) * 2 + pc) * 2 + lh) * 32 + li]);
            }
    auto tile_of = [&](int k) { return decode_item(p, q, j0 + k * nwg_q, 32); };

    // DMA descriptors of this wave's pieces of one 16-channel input stage: (row, col, unit) of the 16-byte unit a lane fetches
    int d_rc[X::NI];
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int U = (k * 4 + wv) * 64 + lane;
        const int row = U / X::RSU, r = U % X::RSU;
        const int cg = r >> 4, r16 = r & 15;
        const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;
        d_rc[k] = row | (col << 8) | ((row < X::IR) ? 1 << 16 : 0) | (w << 17);
    }
    auto fetch_input = [&](int k) {                            // both chunks of tile k's 20x20 halo -> stage buffers 0 / 1
        const bool valid = k < n_mine;
        const Item it = tile_of(valid ? k : 0);
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co;
        const int iy0 = it.ty * X::TH - 2, ix0 = it.tx * X::TW - 2;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int kk = 0; kk < X::NI; ++kk) {
                int rc = d_rc[kk];
                asm volatile("" : "+v"(rc));
                const int row = rc & 255, col = (rc >> 8) & 255, w = (rc >> 17) & 3;
                const int iy = iy0 + row, ix = ix0 + col;
                const int ok = ((rc >> 16) & 1) & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W) & (int)valid;
                const unsigned long long a_in = (unsigned long long)(in + ((iy * p.W + ix) * p.in_cs + ch * 16 + w * 4));
                const unsigned long long a = ok ? a_in : (unsigned long long)p.zero;
                // Inline asm, not __builtin_amdgcn_global_load_lds: hipcc guards every LDS read that follows a DMA it can see with a
                // vmcnt wait (it cannot tell the buffers apart) -- here the scale-table read of step 2, i.e. the halo's whole memory
                // latency in front of conv2.  The fetch is ordered by hand instead: vmcnt(0) + barrier at the end of the tile.
                const unsigned dst = lds0 + (unsigned)(ch * X::STAGE_BYTES + (kk * 4 + wv) * 1024);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(a), "s"(dst) : "memory");      // (m0 is reserved: hipcc keeps nothing in it; no other LDS-DMA / movrel here)
            }
    };
    // conv1 block slots of this wave: blocks wv, wv + 4, wv + 8 of the 11 (slot 2 of wave 3 is idle).  Blocks 0-8: m rows 2b, 2b + 1,
    // m columns 1..16; block 9: m rows 0..15, columns {0, 17}; block 10: rows 16, 17, columns {0, 17} (4 lanes).
    int my[3], mx[3];
    bool act[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        const int b = wv + 4 * sl;
        if (b < 9) { my[sl] = 2 * b + li / 16; mx[sl] = 1 + li % 16; act[sl] = true; }
        else if (b == 9) { my[sl] = li >> 1; mx[sl] = (li & 1) * 17; act[sl] = true; }
        else if (b == 10) { my[sl] = li < 4 ? 16 + (li >> 1) : 0; mx[sl] = li < 4 ? (li & 1) * 17 : 0; act[sl] = li < 4; }
        else { my[sl] = 0; mx[sl] = 0; act[sl] = false; }
    }
    int xa[3][3][2];                                           // conv1: fragment address of input pixel (my, mx + dx), unit 2 lh + pc; + dy rows
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) xa[sl][dx][pc] = (my[sl] * X::RSU + unit_of(mx[sl] + dx, lh * 2 + pc)) * 16;
    int xa2[3][2];                                             // conv2: m pixel (4 wv + li / 16 + dy, li % 16 + dx); block j adds 2 rows
    {
        const int prow = 4 * wv + li / 16, pcol = li % 16;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) xa2[dx][pc] = (prow * X::RSU + unit_of(pcol + dx, lh * 2 + pc)) * 16;
    }
    char* sE = sBuf + X::OFF_E + wv * EPI_WAVE;

    fetch_input(0);
    // A "use" of every weight register in front of the tile loop: hipcc then waits for these loads HERE, once.  Left to the first
    // MFMA inside the loop its wait is a conservative vmcnt(0) on every iteration -- in front of conv1 it would wait out the
    // residual prefetch, in front of conv2 the halo DMA that is meant to fly under steps 3-4 (profiles/r03_h2r_notes.md).
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            asm volatile("" :: "v"(w1[ch][tap][0]), "v"(w1[ch][tap][1]), "v"(w2[ch][tap][0]), "v"(w2[ch][tap][1]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // tile 0's halo and the scale table are in; weights in registers

#pragma unroll 1
    for (int k = 0; k < n_mine; ++k) {
        const Item it = tile_of(k);
        EpiRes<2, 1> pre;                                      // the residual rows of this tile, in the epilogue's ownership
        {
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));
            conv_epilogue_prefetch<3, 1, 2, 1, 16, 16, 4>(p, it, wv, lane_p, pre);
        }
        // ---- 1. conv1: 27 block-steps per chunk, reads PF steps ahead
        f32x16 acc1[3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[sl][r] = 0.f;
        {
            constexpr int PF = 3, NU = 54;                     // unit u: chunk u / 27, tap (u % 27) / 3, slot u % 3
            frag xf[PF + 1][2];
            auto read_x = [&](int u) {
                const int ch = u / 27, tap = (u % 27) / 3, sl = u % 3;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    xf[u % (PF + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + ch * X::STAGE_BYTES + xa[sl][tap % 3][pc] + (tap / 3) * (X::RSU * 16));
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) read_x(u);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int ch = u / 27, tap = (u % 27) / 3, sl = u % 3;
                if (u + PF < NU) read_x(u + PF);
                const frag (&x)[2] = xf[u % (PF + 1)];
                acc1[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ch][tap][1], x[0], acc1[sl], 0, 0, 0);
                acc1[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ch][tap][0], x[1], acc1[sl], 0, 0, 0);
                acc1[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ch][tap][0], x[0], acc1[sl], 0, 0, 0);
                if (u + PF < NU) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
        }
        // ---- 2. every wave is done with the input halo: fetch the next tile's; hand m over
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        fetch_input(k + 1);                                    // (the zero page beyond the last tile: the stream stays branch-free)
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int iy = it.ty * X::TH - 1 + my[sl], ix = it.tx * X::TW - 1 + mx[sl];
            const bool inside = act[sl] && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cl = g4 * 8 + lh * 4;
                const float4 sc = *reinterpret_cast<const float4*>(sS + cl);
                const float4 sh = *reinterpret_cast<const float4*>(sS + 32 + cl);
                float4 v;
                v.x = fmaxf(fmaf(acc1[sl][g4 * 4 + 0], sc.x, sh.x), 0.f);
                v.y = fmaxf(fmaf(acc1[sl][g4 * 4 + 1], sc.y, sh.y), 0.f);
                v.z = fmaxf(fmaf(acc1[sl][g4 * 4 + 2], sc.z, sh.z), 0.f);
                v.w = fmaxf(fmaf(acc1[sl][g4 * 4 + 3], sc.w, sh.w), 0.f);
                if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                h2_pack(v, p.act_scale, hi, lo);
                // channels 8 g4 + 4 lh ..: octet g4 = chunk g4 >> 1, octet-in-chunk g4 & 1; half lh of its high / low unit
                char* m = sM + (g4 >> 1) * X::MID_PLANE + lh * 8;
                if (act[sl]) {
                    *reinterpret_cast<uint2*>(m + (my[sl] * X::RSU + unit_of(mx[sl], (g4 & 1) * 2 + 0)) * 16) = hi;
                    *reinterpret_cast<uint2*>(m + (my[sl] * X::RSU + unit_of(mx[sl], (g4 & 1) * 2 + 1)) * 16) = lo;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // m is complete
        // ---- 3. conv2 from m
        f32x16 acc2[2][1];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][0][r] = 0.f;
        {
            constexpr int PF = 3, NU = 36;                     // unit u: chunk u / 18, tap (u % 18) / 2, block u % 2
            frag xf[PF + 1][2];
            auto read_x = [&](int u) {
                const int ch = u / 18, tap = (u % 18) / 2, j = u % 2;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    xf[u % (PF + 1)][pc] = *reinterpret_cast<const frag*>(sM + ch * X::MID_PLANE + xa2[tap % 3][pc] + (2 * j + tap / 3) * (X::RSU * 16));
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) read_x(u);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int ch = u / 18, tap = (u % 18) / 2, j = u % 2;
                if (u + PF < NU) read_x(u + PF);
                const frag (&x)[2] = xf[u % (PF + 1)];
                acc2[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[ch][tap][1], x[0], acc2[j][0], 0, 0, 0);
                acc2[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[ch][tap][0], x[1], acc2[j][0], 0, 0, 0);
                acc2[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[ch][tap][0], x[0], acc2[j][0], 0, 0, 0);
                if (u + PF < NU) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
        }
        // ---- 4. y tile: bn2 + x + ReLU, H2 stores (the ordinary fused epilogue, residual already in registers)
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            conv_epilogue<3, 1, 2, 1, 16, 16, 4>(p, it, acc2, sS + 64, sE, wv, lane_e & 31, lane_e >> 5, pre, true);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next halo has landed (and this tile's stores are out)
        __builtin_amdgcn_s_barrier();                          // ... for every wave; m may be overwritten
    }
}

// `op` is the block's SECOND conv (its residual is the block input x, its output y); `op1` the first (weights / scale / shift).
int launch_bblock32(const romp_op& op1, const romp_op& op, const float* x, float* y, int B, int* queue, hipStream_t st) {
    ROMP_REQUIRE(op.ksize == 3 && op.stride == 1 && op.Cin == 32 && op.Cout == 32 && op.cin_pad == 32 && op.cout_pad == 32 && op.groups == 1 &&
                 op1.ksize == 3 && op1.stride == 1 && op1.Cin == 32 && op1.Cout == 32 && op1.cin_pad == 32 && op1.cout_pad == 32 && op1.groups == 1,
                 "bblock32: two 3x3 stride-1 32 -> 32 convs expected");
    ROMP_REQUIRE(op1.weight_h2 && op1.scale_h2 && op.weight_h2 && op.scale_h2 && op1.relu && op.relu, "bblock32: f16x2 weights and ReLUs expected");
    ROMP_REQUIRE(op1.in_fmt == ROMP_FMT_H2 && op.res_fmt == ROMP_FMT_H2 && op.out_fmt == ROMP_FMT_H2 && op1.act_shift == op.act_shift,
                 "bblock32: H2 tensors expected");
    ROMP_REQUIRE(op.H % 16 == 0 && op.W % 16 == 0 && op1.H == op.H && op1.W == op.W, "bblock32: %dx%d is not a multiple of the 16x16 tile", op.H, op.W);
    ROMP_REQUIRE(op1.in_cstride == op.res_cstride && op1.in_coff == op.res_coff && ((op1.in_cstride | op1.in_coff | op.out_cstride | op.out_coff) & 7) == 0,
                 "bblock32: the residual must be the block input, octet aligned");
    static bool attr = false;
    static float* zero = nullptr;                              // 256 bytes of zeros: what out-of-image lanes of the halo DMA fetch
    static int num_cu = 256;
    if (!attr) {                                               // (romp_net_create calls this path's setup outside any stream capture: bblock_init)
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bblock32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BCfg::LDS_BYTES));
        ROMP_HIP_CHECK(hipMalloc((void**)&zero, 256));
        ROMP_HIP_CHECK(hipMemset(zero, 0, 256));
        int dev = 0;
        hipDeviceProp_t prop;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr = true;
    }
    if (x == nullptr && y == nullptr) return ROMP_OK;          // set-up only
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = x; p.res = x; p.out = y;
    p.w3 = reinterpret_cast<const uint4*>(op1.weight_h2);
    p.wh = reinterpret_cast<const uint4*>(op.weight_h2);
    p.scale = op1.scale_h2; p.w = op1.shift;                   // conv1's epilogue scale / shift (see the kernel's header)
    p.scale_h = op.scale_h2; p.shift = op.shift;
    p.zero = zero;
    p.act_scale = ldexpf(1.f, op.act_shift);
    p.inv_act_scale = ldexpf(1.f, -op.act_shift);
    p.in_h2 = p.out_h2 = p.res_h2 = 1;
    p.queue = queue;
    p.H = p.Ho = op.H; p.W = p.Wo = op.W;
    p.Cout = 32; p.cin_valid = 32; p.cin_pad = 32; p.cout_pad = 32;
    p.in_cs = op1.in_cstride; p.in_co = op1.in_coff; p.in_gs = 0;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff; p.out_gs = 0;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff; p.res_gs = 0;
    p.relu = 1;
    p.tiles_x = op.W / 16; p.tiles_y = op.H / 16; p.tiles_total = B * p.tiles_x * p.tiles_y;
    p.nslices = p.ns_total = 1;
    p.n_queues = (p.tiles_total % 8 == 0) ? 8 : 1;
    p.per_queue = p.tiles_total / p.n_queues;
    p.tile_contig = 1;
    p.vec_io = 1;
    p.pad_h = p.pad_w = 1;
    p.out_rs = op.out_rstride > 0 ? op.out_rstride : p.Wo * op.out_cstride;
    p.out_bs = op.out_bstride > 0 ? op.out_bstride : p.Ho * p.Wo * op.out_cstride;
    long grid = num_cu;                                        // one workgroup per CU
    if (grid > p.tiles_total) grid = p.tiles_total;
    if (p.n_queues == 8) grid = grid >= 8 ? (grid / 8) * 8 : 8;
    hipLaunchKernelGGL(bblock32_kernel, dim3((unsigned)grid), dim3(256), BCfg::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp

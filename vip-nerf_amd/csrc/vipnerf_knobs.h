// Every build-time switch of libvipnerf_hip.so, in ONE place, with its default -- and the string vipnerf_build_info() returns, so that a
// loaded library can say how it was built (include/vipnerf_hip.h).  Override with -DVN_X=v through VIPNERF_EXTRA_FLAGS (build.sh); the
// switch's meaning and the measurements behind its default are documented where it is used (grep the name).
//
// VN_EXP is different in kind: it selects TIMING-ONLY experiment builds that leave out stores, encodings or MFMAs (results are garbage;
// tools/build_exp.sh).  A library built with it reports "VN_EXP=<n>" here, _lib.load() warns, and bench.py / __graft_entry__.smoke()
// refuse to run on it.
#pragma once

#define VN_KNOB_LIST(X)                                                                                                                   \
    /* lane layouts chosen when vipnerf_config.bf16_layout = DEFAULT (vipnerf_api.hip) */                                                \
    X(VN_BF16_NARROW_DEFAULT) X(VN_FP32_NARROW_DEFAULT)                                                                                   \
    /* fp32 / split-precision MLP kernels (vipnerf_common.h, vipnerf_bf16.h, vipnerf_bf16n.h, vipnerf_mlp_*_bf16n.hip) */                 \
    X(VN_STAGE_CHUNKS) X(VN_SPLIT_FMA_MIX) X(VN_INTERLEAVE) X(VN_SKEW) X(VN_DMA_MODE) X(VN_DMA_ISSUERS) X(VN_F16_PRESPLIT) X(VN_DEFER_STORES)             \
    X(VN_STORE_GROUP_A) X(VN_STORE_GROUP_B) X(VN_F32_PERSISTENT) X(VN_F32_EVAL_ROTATE) X(VN_F32B_CONV_GROUP) X(VN_DMA_ROT_WAVES) X(VN_F32_DMA_ISSUERS)                                                              \
    /* single-MFMA 16-bit modes: storage and the two-point-tile kernels (vipnerf_bf16n.h, vipnerf_mlp_pt2.h, vipnerf_mlp_*_pt2.hip) */    \
    X(VN_BF16_H16) X(VN_T16) X(VN_T16_X4) X(VN_T16_NT) X(VN_PT2_SPREAD) X(VN_PT2_SKEW) X(VN_PT2_G) X(VN_PT2_D)                  \
    X(VN_PT2_TRAIN_KEEP) X(VN_PT2_EVAL_KEEP) X(VN_PT2_FAST_PE)                                                                                              \
    /* weight gradients (vipnerf_wgrad.hip, vipnerf_wgrad16.hip) */                                                                       \
    X(VN_WGRAD_SIGMA_FUSED) X(VN_WGRAD_VIEW_FUSED) X(VN_WGRAD_HEADS_FUSED) X(VN_WGRAD_LATE_LOAD) X(VN_WGRAD_LATE_STORE) X(VN_WGRAD_STORE_SKEW) X(VN_WGRAD_FAST) X(VN_WGRAD_BIAS_WK0) X(VN_WGRAD_VECFRAG) X(VN_WGRAD_PREFETCH) X(VN_WGRAD_DMA) X(VN_WGRAD_W8) X(VN_WGRAD_PIPE) X(VN_WGRAD_ONE_ROUND) X(VN_WGRAD_ROUNDS) X(VN_WG16_BIG_WM) X(VN_WG16_BIG_WN)        \
    X(VN_WG16_BIG_NB) X(VN_WG16_HYBRID) X(VN_WG16_SIGMA_FUSED) X(VN_WG16_DMA_PIECES) X(VN_WG16_THIN_HYBRID) X(VN_WG16_BIG_SLOTS) X(VN_WG16_VIEW_FUSED) X(VN_WG16_VIEW_WN)          \
    /* optimizer (vipnerf_api.hip) */                                                                                                     \
    X(VN_ADAM_FMA_MASK)

#ifndef VN_BF16_NARROW_DEFAULT
#define VN_BF16_NARROW_DEFAULT 1
#endif
#ifndef VN_FP32_NARROW_DEFAULT
#define VN_FP32_NARROW_DEFAULT 1
#endif
#ifndef VN_STAGE_CHUNKS
#define VN_STAGE_CHUNKS 64
#endif
#ifndef VN_SPLIT_FMA_MIX
#define VN_SPLIT_FMA_MIX 1
#endif
#ifndef VN_INTERLEAVE
#define VN_INTERLEAVE 1
#endif
#ifndef VN_SKEW
#define VN_SKEW 0
#endif
#ifndef VN_DMA_MODE
#define VN_DMA_MODE 1            // 1: one wave issues a whole stage (ROTATE); 2: every wave its share, staggered over the stage
#endif
#ifndef VN_DMA_ISSUERS
#define VN_DMA_ISSUERS 1         // VN_DMA_MODE 1: waves that share a stage's DMA (1, 2, 4 or 8; vipnerf_bf16.h WStreamT)
#endif
#ifndef VN_F16_PRESPLIT
#define VN_F16_PRESPLIT 1
#endif
#ifndef VN_DEFER_STORES
#define VN_DEFER_STORES 1
#endif
#ifndef VN_STORE_GROUP_A
#define VN_STORE_GROUP_A 6
#endif
#ifndef VN_STORE_GROUP_B
#define VN_STORE_GROUP_B 12
#endif
#ifndef VN_F32_PERSISTENT
#define VN_F32_PERSISTENT 1       // exact-fp32 data-gradient kernel: persistent workgroups (one per CU, tiles round robin, the weight stream continuous across tiles)
#endif
#ifndef VN_F32_EVAL_ROTATE
#define VN_F32_EVAL_ROTATE 1      // exact-fp32 EVAL kernel: 1 = the training kernels' weight stream (the four older waves issue a quarter of a stage each: 0.908 -> 0.918 of the peak), 0 = every wave its eighth
#endif
#ifndef VN_WGRAD_SIGMA_FUSED
#define VN_WGRAD_SIGMA_FUSED 1    // exact fp32: the sigma head's weight gradient as weighted column sums inside the feature layer's 256 x 256 GEMM (k_wgrad256_w8)
#endif
#ifndef VN_WGRAD_VIEW_FUSED
#define VN_WGRAD_VIEW_FUSED 1     // exact fp32: the view layer's 128 x 256 and per-direction 128 x 32 weight-gradient GEMMs in one launch over dYv_0..V (k_wgrad_view)
#endif
#ifndef VN_F32_DMA_ISSUERS
#define VN_F32_DMA_ISSUERS 4      // exact-fp32 MLP kernels: older waves (0..3) that share a stage's DMA (1, 2 or 4)
#endif
#ifndef VN_DMA_ROT_WAVES
#define VN_DMA_ROT_WAVES 4       // VN_DMA_MODE 1: the issuer of a stage's DMA rotates over waves 0 .. n - 1 (4: the older wave of each SIMD; 8: every wave)
#endif
#ifndef VN_F32B_CONV_GROUP
#define VN_F32B_CONV_GROUP 2     // k_mlp_bwd_f32: MFMA group of a stage behind which the next stage's operand k-steps are converted
#endif
#ifndef VN_BF16_H16
#define VN_BF16_H16 1
#endif
#ifndef VN_T16
#define VN_T16 1
#endif
#ifndef VN_T16_X4
#define VN_T16_X4 1
#endif
#ifndef VN_T16_NT
#define VN_T16_NT 1              // nontemporal tile stores (the data is next read by another kernel, GBs later); 0: plain stores
#endif
#ifndef VN_PT2_SPREAD
#define VN_PT2_SPREAD 4
#endif
#ifndef VN_PT2_SKEW
#define VN_PT2_SKEW 0            // 1: waves 4..7 (the second wave of every SIMD) send a stage's deferred stores half a part later than waves 0..3
#endif
#ifndef VN_PT2_G
#define VN_PT2_G 1
#endif
#ifndef VN_PT2_D
#define VN_PT2_D 2
#endif
#ifndef VN_PT2_TRAIN_KEEP
#define VN_PT2_TRAIN_KEEP 0      // training: 0 = gamma(x)'s fragments reloaded from the activation store at layer 5; 1 = kept in registers (more spills: forward 1.55 -> 1.67 ms per step, not kept)
#endif
#ifndef VN_PT2_EVAL_KEEP
#define VN_PT2_EVAL_KEEP 1       // eval: 1 = gamma(x)'s fragments stay in 16 registers from layer 0 to layer 5 (measured: fp16 1023 -> 1080, bf16 1181 -> 1226 TFLOP/s); 0 = evaluated again at layer 5
#endif
#ifndef VN_PT2_FAST_PE
#define VN_PT2_FAST_PE 1         // single-MFMA 16-bit kernels: gamma(x), gamma(dir) with v_fract + v_sin_f32 / v_cos_f32 (vipnerf_bf16n.h sincos_rev); 0: sincosf
#endif
#ifndef VN_WGRAD_LATE_LOAD
#define VN_WGRAD_LATE_LOAD 2       // exact-fp32 256 x 256 weight gradients: the k-step (of 16) behind which a wave issues the next block's global loads (a SIMD's second wave: 4 steps later); -1: at the block's top, stores at its end
#endif
#ifndef VN_WGRAD_LATE_STORE
#define VN_WGRAD_LATE_STORE 12     // ... and behind which it stores them to LDS (a SIMD's second wave: 2 steps later)
#endif
#ifndef VN_WGRAD_HEADS_FUSED
#define VN_WGRAD_HEADS_FUSED 1     // exact fp32, V <= 1: the output head's weight gradient rides in k_wgrad_view (the waves without a direction tile take it); 0: its own launch (k_wgrad<1,1,1>)
#endif
#ifndef VN_WGRAD_STORE_SKEW
#define VN_WGRAD_STORE_SKEW 2      // ... the second wave's lag in k-steps for those LDS stores
#endif
#ifndef VN_WGRAD_FAST
#define VN_WGRAD_FAST 1            // k_wgrad256_w8: a block loop without range logic for whole [P][256] operands and whole 32-point blocks (8.15 -> 7.90 ms per step with the operand bases pinned in SGPRs)
#endif
#ifndef VN_WGRAD_BIAS_WK0
#define VN_WGRAD_BIAS_WK0 1        // k_wgrad256_w8: only the waves that store a bias sum (the older wave of every SIMD) form it, behind a scalar branch at the block's end (7.89 -> 7.85)
#endif
#ifndef VN_WGRAD_VECFRAG
#define VN_WGRAD_VECFRAG 1         // k_wgrad256_w8: a wave's tiles are interleaved feature sets, its fragments one 8-byte + one 16-byte LDS read per k-step at immediate offsets (no address arithmetic among the MFMAs)
#endif
#ifndef VN_WGRAD_PREFETCH
#define VN_WGRAD_PREFETCH 1        // ... and k-step s + 1's fragments are requested before k-step s's MFMAs (7.86 -> 7.79)
#endif
#ifndef VN_WGRAD_DMA
#define VN_WGRAD_DMA 0           // exact-fp32 256 x 256 weight gradients: 1 = operand blocks HBM -> LDS by DMA instead of through registers -- built, correct, and measured SLOWER (9.10 vs 8.22 ms per step: docs/HISTORY.md 5); off
#endif
#ifndef VN_WGRAD_W8
#define VN_WGRAD_W8 2            // exact-fp32 256 x 256 weight gradients: 0 = the 4-wave k_wgrad<2,8,4>; 2 / 4 = k_wgrad256_w8 with 8 / 16 waves
#endif
#ifndef VN_WGRAD_PIPE
#define VN_WGRAD_PIPE 1
#endif
#ifndef VN_WGRAD_ONE_ROUND
#define VN_WGRAD_ONE_ROUND 1     // point chunks: one round of workgroups per launch where the level is large enough (like vipnerf_wgrad16.hip)
#endif
#ifndef VN_WGRAD_ROUNDS
#define VN_WGRAD_ROUNDS 1
#endif
#ifndef VN_WG16_BIG_WM
#define VN_WG16_BIG_WM 2         // wave grid and ring depth of the 256 x 256 kernel
#endif
#ifndef VN_WG16_BIG_WN
#define VN_WG16_BIG_WN 4
#endif
#ifndef VN_WG16_BIG_NB
#define VN_WG16_BIG_NB 4
#endif
#ifndef VN_WG16_HYBRID
#define VN_WG16_HYBRID 1         // the 256 x 256 kernel streams half of each block by DMA, half through registers (k_wg16's HY)
#endif
#ifndef VN_WG16_SIGMA_FUSED
#define VN_WG16_SIGMA_FUSED 1    // the sigma head rides in the feature layer's GEMM (XA); 0: its own 16 x 256 launch
#endif
#ifndef VN_WG16_DMA_PIECES
#define VN_WG16_DMA_PIECES 2     // of a wave's 4 pieces per block (256 x 256 launch): measured 1.36 ms per 4096-ray step; see docs/HISTORY.md 4.3a for 0 / 1 / 3
#endif
#ifndef VN_WG16_THIN_HYBRID
#define VN_WG16_THIN_HYBRID 0    // the 256 x 64 and 128 x 256 launches on the hybrid trip stream too (measured: docs/HISTORY.md 4.3a)
#endif
#ifndef VN_WG16_VIEW_FUSED
#define VN_WG16_VIEW_FUSED 1     // 16-bit modes: the view layer's 128 x 256 and per-direction 128 x 32 weight-gradient GEMMs in one launch over dYv_0..V (k_wg16_view); the data-gradient kernels then write no dYvsum
#endif
#ifndef VN_WG16_VIEW_WN
#define VN_WG16_VIEW_WN 2        // k_wg16_view: 2 x WN waves (2: four waves own 4 x 8 tiles each; 4: eight waves own 4 x 4)
#endif
#ifndef VN_WG16_BIG_SLOTS
#define VN_WG16_BIG_SLOTS 256    // workgroups of the 256 x 256 launch at a large level: one round of the chip (two / three rounds measured: docs/HISTORY.md 4.3a)
#endif
#ifndef VN_ADAM_FMA_MASK
#define VN_ADAM_FMA_MASK 7       // which of torch's three update expressions its kernels contract into an fma on gfx950 (tests/test_hip_fullsize.py)
#endif

#define VN_KNOB_STR2(x) #x
#define VN_KNOB_STR(x) VN_KNOB_STR2(x)
#define VN_KNOB_ITEM(name) " " #name "=" VN_KNOB_STR(name)
#if defined(VN_EXP)
#define VN_EXP_STR "VN_EXP=" VN_KNOB_STR(VN_EXP)
#define VN_EXP_VALUE (VN_EXP)
#else
#define VN_EXP_STR "VN_EXP=unset"
#define VN_EXP_VALUE (-1)
#endif
// "libvipnerf_hip abi=<n> arch=gfx950 VN_EXP=unset VN_BF16_NARROW_DEFAULT=1 ..."
#define VN_BUILD_INFO_KNOBS VN_EXP_STR VN_KNOB_LIST(VN_KNOB_ITEM)

// Dispatch table of the implicit-GEMM convolution instantiations (see conv_igemm.h).
#pragma once
#include "conv_igemm.h"

// tile-shape ids: chosen from the logical output width
enum { SHAPE_W128 = 0, SHAPE_W64 = 1, SHAPE_W32 = 2, SHAPE_W16 = 3, SHAPE_W8 = 4, N_SHAPES = 5 };
// block configs: A = 128 co x 128 pos (waves 2x2, wave tile 64x64); B = 64 co x 128 pos (waves 1x4, wave tile 64x32);
// C = 32 co x 128 pos (waves 1x4, wave tile 32x32); D = 64 co x 256 pos (waves 1x4, wave tile 64x64): per MFMA half the
// weight-tile traffic of A and B and 25-33 % less patch traffic (a 2x128 / 4x64 / 8x32 position tile has less halo per
// position than 1x128 / 2x64 / 4x32); 2-D 3x3 layers only
// E = 64 co x 512 pos (waves 1x4, wave tile 64x128 = 8 accumulators): half of D's staging traffic per MFMA again, 2 waves / SIMD
// F = 32 co x 256 pos (waves 1x4, wave tile 32x64): D for layers with 32 or fewer output channels
// G = 128 co x 256 pos (waves 2x2, wave tile 64x128 = 8 accumulators), fp16-operand 3x3 kernel only: the patch staged per block
//     feeds twice the MFMAs of D and a step reads 6 fragments for 8 MFMAs instead of 4 for 4; one block per CU (512 registers)
enum { CFG_A = 0, CFG_B = 1, CFG_C = 2, CFG_D = 3, CFG_E = 4, CFG_F = 5, CFG_G = 6, N_CFGS = 7 };

typedef int (*conv_launch_fn)(ConvArgs, hipStream_t);


#ifndef EMO_CONV_KC_3X3
#define EMO_CONV_KC_3X3 4
#endif
#ifndef EMO_CONV_KC_3X3_A
#define EMO_CONV_KC_3X3_A EMO_CONV_KC_3X3   /* the 128-row config may use a shorter stage (smaller LDS tiles -> 4 instead of 3 blocks/CU) */
#endif
#define EMO_CONV_KC_1X1 16   /* 32 measured slower (64 KiB+ LDS, 163 VGPR: 67 vs 73 TF on 1536->512 @64^2) */
#define EMO_CONV_KC_1X7 4

#define EMO_CONV_KC_F16_3X3 16   /* fp16-operand kernels (conv_igemm_f16.h): input channels per stage, a multiple of 16 */
#define EMO_CONV_KC_F16_1X1 32

// fp16-operand kernels (conv_igemm_f16.h): 64 x 256 tiles only (widths that are multiples of 128, or 64 / 32)
#define CONV_FOR_SHAPE_F16(KH, KW, KC, shape, ups)                                                              \
  ((shape) == SHAPE_W128 ? ((ups) ? &conv_igemm_f16_launch<KH, KW, KC, 1, 2, 128, 2, 2, 1, 4, true>              \
                                  : &conv_igemm_f16_launch<KH, KW, KC, 1, 2, 128, 2, 2, 1, 4, false>)            \
   : (shape) == SHAPE_W64 ? ((ups) ? &conv_igemm_f16_launch<KH, KW, KC, 1, 4, 64, 2, 2, 1, 4, true>              \
                                   : &conv_igemm_f16_launch<KH, KW, KC, 1, 4, 64, 2, 2, 1, 4, false>)            \
   : (shape) == SHAPE_W32 ? ((ups) ? &conv_igemm_f16_launch<KH, KW, KC, 1, 8, 32, 2, 2, 1, 4, true>              \
                                   : &conv_igemm_f16_launch<KH, KW, KC, 1, 8, 32, 2, 2, 1, 4, false>)            \
                          : (conv_launch_fn) nullptr)

// fp16-operand kernels, 128 x 256 tiles (CFG_G)
#define CONV_FOR_SHAPE_F16_G(KH, KW, KC, shape, ups)                                                            \
  ((shape) == SHAPE_W128 ? ((ups) ? &conv_igemm_f16_launch<KH, KW, KC, 1, 2, 128, 2, 4, 2, 2, true>              \
                                  : &conv_igemm_f16_launch<KH, KW, KC, 1, 2, 128, 2, 4, 2, 2, false>)            \
   : (shape) == SHAPE_W64 ? ((ups) ? &conv_igemm_f16_launch<KH, KW, KC, 1, 4, 64, 2, 4, 2, 2, true>              \
                                   : &conv_igemm_f16_launch<KH, KW, KC, 1, 4, 64, 2, 4, 2, 2, false>)            \
   : (shape) == SHAPE_W32 ? ((ups) ? &conv_igemm_f16_launch<KH, KW, KC, 1, 8, 32, 2, 4, 2, 2, true>              \
                                   : &conv_igemm_f16_launch<KH, KW, KC, 1, 8, 32, 2, 4, 2, 2, false>)            \
                          : (conv_launch_fn) nullptr)

#define CONV_FOR_SHAPE(KH, KW, KC, TM, TP, WGM, WGP, shape, ups)                                              \
  ((shape) == SHAPE_W128 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 1, 128, TM, TP, WGM, WGP, true>          \
                                  : &conv_igemm_launch<KH, KW, KC, 1, 1, 128, TM, TP, WGM, WGP, false>)        \
   : (shape) == SHAPE_W64 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 2, 64, TM, TP, WGM, WGP, true>          \
                                   : &conv_igemm_launch<KH, KW, KC, 1, 2, 64, TM, TP, WGM, WGP, false>)        \
   : (shape) == SHAPE_W32 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 4, 32, TM, TP, WGM, WGP, true>          \
                                   : &conv_igemm_launch<KH, KW, KC, 1, 4, 32, TM, TP, WGM, WGP, false>)        \
   : (ups)               ? (conv_launch_fn) nullptr                                                            \
   : (shape) == SHAPE_W16 ? &conv_igemm_launch<KH, KW, KC, 1, 8, 16, TM, TP, WGM, WGP, false>                 \
   : (shape) == SHAPE_W8  ? &conv_igemm_launch<KH, KW, KC, 2, 8, 8, TM, TP, WGM, WGP, false>                  \
                          : (conv_launch_fn) nullptr)

// 64 x 256 tiles (CFG_D): 2-D widths that are multiples of 128, or 64 / 32
#define CONV_FOR_SHAPE_D(KH, KW, KC, shape, ups)                                                              \
  ((shape) == SHAPE_W128 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 2, 128, 2, 2, 1, 4, true>                \
                                  : &conv_igemm_launch<KH, KW, KC, 1, 2, 128, 2, 2, 1, 4, false>)              \
   : (shape) == SHAPE_W64 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 4, 64, 2, 2, 1, 4, true>                \
                                   : &conv_igemm_launch<KH, KW, KC, 1, 4, 64, 2, 2, 1, 4, false>)              \
   : (shape) == SHAPE_W32 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 8, 32, 2, 2, 1, 4, true>                \
                                   : &conv_igemm_launch<KH, KW, KC, 1, 8, 32, 2, 2, 1, 4, false>)              \
                          : (conv_launch_fn) nullptr)

// 64 x 512 tiles (CFG_E): 2-D widths that are multiples of 128, or 64
#define CONV_FOR_SHAPE_E(KH, KW, KC, shape, ups)                                                              \
  ((shape) == SHAPE_W128 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 4, 128, 2, 4, 1, 4, true>                \
                                  : &conv_igemm_launch<KH, KW, KC, 1, 4, 128, 2, 4, 1, 4, false>)              \
   : (shape) == SHAPE_W64 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 8, 64, 2, 4, 1, 4, true>                \
                                   : &conv_igemm_launch<KH, KW, KC, 1, 8, 64, 2, 4, 1, 4, false>)              \
                          : (conv_launch_fn) nullptr)

// 32 x 256 tiles (CFG_F): widths that are multiples of 128, or 64 / 32
#define CONV_FOR_SHAPE_F(KH, KW, KC, shape, ups)                                                              \
  ((shape) == SHAPE_W128 ? ((ups) ? &conv_igemm_launch<KH, KW, KC, 1, 2, 128, 1, 2, 1, 4, true>                \
                                  : &conv_igemm_launch<KH, KW, KC, 1, 2, 128, 1, 2, 1, 4, false>)              \
   : (ups)               ? (conv_launch_fn) nullptr                                                            \
   : (shape) == SHAPE_W64 ? &conv_igemm_launch<KH, KW, KC, 1, 4, 64, 1, 2, 1, 4, false>                        \
   : (shape) == SHAPE_W32 ? &conv_igemm_launch<KH, KW, KC, 1, 8, 32, 1, 2, 1, 4, false>                        \
                          : (conv_launch_fn) nullptr)

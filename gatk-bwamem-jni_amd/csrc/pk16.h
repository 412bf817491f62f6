// pk16.h -- two 16-bit integers per 32-bit register (the v_pk_*_i16 / _u16 instructions of gfx9+).
//
// Used by the packed form of the extension DP (k_extend.hip: extend_wave_pk2: two 64-column chunks of a DP row share one
// instruction stream, the low half of every register holding a column of the first chunk and the high half the same lane's
// column of the second) and by the packed form of mate rescue's local alignment (sw_common.h: sw_core_wave8_u8: two alignments
// per lane).  Three builds see this header: the device pass of hipcc (clang vector types, plus a few instructions written as
// `asm` because the optimiser rewrites their vector-IR forms into per-half compares and selects), the host pass of hipcc (the same
// vector types, no asm: never executed), and the g++ build of the test emulation (tests/emu), which computes the same per half.
#pragma once
#include <stdint.h>
#include "dev_common.h"

#if defined(__clang__)
typedef short pk_s16_t __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16_t __attribute__((ext_vector_type(2)));
DEV uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (pk_s16_t)(__builtin_bit_cast(pk_s16_t, a) + __builtin_bit_cast(pk_s16_t, b))); }
DEV uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (pk_s16_t)(__builtin_bit_cast(pk_s16_t, a) - __builtin_bit_cast(pk_s16_t, b))); }
DEV uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_s16_t, a), __builtin_bit_cast(pk_s16_t, b))); }
DEV uint32_t pk_minu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pk_u16_t, a), __builtin_bit_cast(pk_u16_t, b))); }
#if defined(__HIP_DEVICE_COMPILE__)
// 0xffff where the half is negative (op_sel_hi:[0,1]: both halves take the shift count from the low half of the inline constant)
DEV uint32_t pk_sra15(uint32_t a) { uint32_t r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
// unsigned a - b per half, saturating at 0 (the clamp bit of the packed subtraction)
DEV uint32_t pk_subs(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
DEV uint32_t pk_sra15(uint32_t a) { return __builtin_bit_cast(uint32_t, (pk_s16_t)(__builtin_bit_cast(pk_s16_t, a) >> (pk_s16_t)15)); }
DEV uint32_t pk_subs(uint32_t a, uint32_t b) { const pk_u16_t x = __builtin_bit_cast(pk_u16_t, a), y = __builtin_bit_cast(pk_u16_t, b); return __builtin_bit_cast(uint32_t, (pk_u16_t)(__builtin_elementwise_max(x, y) - y)); }
#endif
// one byte of lo and one of hi as two unsigned 16-bit halves (v_perm_b32): byte klo of lo -> low half, byte khi of hi -> high half
DEV uint32_t pk_bytes2(uint32_t lo, uint32_t hi, uint32_t klo, uint32_t khi) { return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (4u + khi) << 16 | klo); }
#else
DEV uint32_t pk_join(int lo, int hi) { return (uint32_t)(uint16_t)lo | (uint32_t)(uint16_t)hi << 16; }
DEV int pk_lo_(uint32_t a) { return (int16_t)(a & 0xffffu); }
DEV int pk_hi_(uint32_t a) { return (int16_t)(a >> 16); }
DEV uint32_t pk_add(uint32_t a, uint32_t b) { return pk_join(pk_lo_(a) + pk_lo_(b), pk_hi_(a) + pk_hi_(b)); }
DEV uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_join(pk_lo_(a) - pk_lo_(b), pk_hi_(a) - pk_hi_(b)); }
DEV uint32_t pk_max(uint32_t a, uint32_t b) { return pk_join(pk_lo_(a) > pk_lo_(b) ? pk_lo_(a) : pk_lo_(b), pk_hi_(a) > pk_hi_(b) ? pk_hi_(a) : pk_hi_(b)); }
DEV uint32_t pk_minu(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al < bl ? al : bl) | (ah < bh ? ah : bh) << 16; }
DEV uint32_t pk_sra15(uint32_t a) { return (a & 0x8000u ? 0xffffu : 0u) | (a & 0x80000000u ? 0xffff0000u : 0u); }
DEV uint32_t pk_subs(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al > bl ? al - bl : 0u) | (ah > bh ? ah - bh : 0u) << 16; }
DEV uint32_t pk_bytes2(uint32_t lo, uint32_t hi, uint32_t klo, uint32_t khi) { return (lo >> (8 * klo) & 0xffu) | (hi >> (8 * khi) & 0xffu) << 16; }
#endif
// ---- the same in every build: pieces the banded global alignment (k_cigar.hip: global_wave_diag_pk) adds
#if defined(__HIP_DEVICE_COMPILE__)
DEV uint32_t pk_maxu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_u16_t, a), __builtin_bit_cast(pk_u16_t, b))); }
// signed a - b / a + b per half, saturating at -32768 / 32767 (the clamp bit)
DEV uint32_t pk_subss(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV uint32_t pk_addss(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
// bytes 0 and 2 of the result picked from the eight bytes {s1 (selectors 0..3), s0 (4..7)} by bytes 0 and 2 of sel; bytes 1 and 3 of sel are 0x0c (zero)
DEV uint32_t pk_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
#else
DEV uint32_t pk_maxu(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al > bl ? al : bl) | (ah > bh ? ah : bh) << 16; }
DEV int pk_sat16_(int x) { return x < -32768 ? -32768 : x > 32767 ? 32767 : x; }
DEV uint32_t pk_subss(uint32_t a, uint32_t b) { return (uint32_t)(uint16_t)pk_sat16_((int)(int16_t)(a & 0xffffu) - (int)(int16_t)(b & 0xffffu)) | (uint32_t)(uint16_t)pk_sat16_((int)(int16_t)(a >> 16) - (int)(int16_t)(b >> 16)) << 16; }
DEV uint32_t pk_addss(uint32_t a, uint32_t b) { return (uint32_t)(uint16_t)pk_sat16_((int)(int16_t)(a & 0xffffu) + (int)(int16_t)(b & 0xffffu)) | (uint32_t)(uint16_t)pk_sat16_((int)(int16_t)(a >> 16) + (int)(int16_t)(b >> 16)) << 16; }
DEV uint32_t pk_perm_byte_(uint32_t s0, uint32_t s1, uint32_t k) { return k < 4 ? (s1 >> (8 * k)) & 0xffu : k < 8 ? (s0 >> (8 * (k - 4))) & 0xffu : 0u; }
DEV uint32_t pk_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return pk_perm_byte_(s0, s1, sel & 0xffu) | pk_perm_byte_(s0, s1, (sel >> 16) & 0xffu) << 16; }
#endif
// signed value of one half
DEV int pk_shalf(uint32_t a, int h) { return (int)(int16_t)(h ? a >> 16 : a & 0xffffu); }
// byte k of lo and of hi (k = 0..3)
DEV uint32_t pk_bytes(uint32_t lo, uint32_t hi, int k) { return pk_bytes2(lo, hi, (uint32_t)k, (uint32_t)k); }
// both halves = the 16 low bits of x / the two halves from two ints
DEV uint32_t pk_both(int x) { return (uint32_t)(uint16_t)x * 0x10001u; }
DEV uint32_t pk_pair(int lo, int hi) { return (uint32_t)(uint16_t)lo | (uint32_t)(uint16_t)hi << 16; }
DEV int pk_half(uint32_t a, int h) { return (int)(h ? a >> 16 : a & 0xffffu); }           // unsigned value of one half

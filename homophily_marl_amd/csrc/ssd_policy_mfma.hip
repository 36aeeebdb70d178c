// ssd_policy_mfma.hip -- the rollout-time controller on the 16-bit matrix cores (include/ssd_hip.h: ssd_policy_encode,
// ssd_policy_head_env / _inc, ssd_policy_pack_*).
//
// What is evaluated (reference): HomophilyAgent.rgb_preprocess (homophily_agent.py:20-27,213-214), the _build_inputs tail
// (homophily_controller.py:137-184), forward_env / forward_inc (homophily_agent.py:154-208: fc1 -> LeakyReLU -> hand-written GRU
// cell -> dueling Q) and the epsilon-greedy selector (action_selectors.py:44-68).
//
// Arithmetic.  The reference computes in f32.  gfx950's f32-input MFMA runs at the VALU rate (1/16 of the f16 / bf16 rate), so
// every f32 product is evaluated on v_mfma_f32_16x16x32_f16 from two-term splits x = hi + lo (hi = f16(x), lo = f16(x - hi): 22
// significand bits) as lo_w*hi_x + hi_w*lo_x + hi_w*hi_x with f32 accumulation: three MFMAs at 16x the f32 rate.  Exact powers of
// two scale weights and activations so that the lo terms stay in f16's normal range; they are divided out of the f32 result.
// PREC = 1 is the reduced-precision variant: single bf16 terms, one MFMA per product.
//
// Layout.  Every product is computed transposed, D^T = W^T X^T: the weight fragment is the A operand (lane (q, m): output
// feature m of a 16-feature tile, 8 reduction indices of quarter q), 16 activation rows are the B operand (lane (q, m): row m),
// and D puts the activation ROW on the lane (m) and 4 consecutive OUTPUT features (4 q + r) in the lane's registers.  With the
// reduction index of K-step s, quarter q, element j chosen as k = 32 s + 16 (j >> 2) + 4 q + (j & 3), the result tiles 2 s and
// 2 s + 1 of one product are, element for element, the B operand of K-step s of the next product: fc1 -> GRU -> dueling (and
// conv -> Linear in the encoder) chain in registers without LDS round trips or lane shuffles, and the GRU gate arithmetic is
// lane-local.  Weight fragments are pre-swizzled by the pack kernels below so that one ds_read_b128 / global_load_dwordx4 per
// lane fetches a whole A operand.
#include "ssd_policy_common.h"

namespace ssd {
#define SSD_GLOBAL __attribute__((address_space(1)))    // a pointer type that carries the global address space


using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using b8 = __attribute__((ext_vector_type(8))) __bf16;

template <int PREC>
__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (PREC == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// 8 f32 values -> the 16-bit fragment(s): hi (and, PREC 2, lo = v - hi; the subtraction is exact)
template <int PREC>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    if constexpr (PREC == 2) {
        h8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[j] = (_Float16)v[j]; l[j] = (_Float16)(v[j] - (float)h[j]); }
        hi = __builtin_bit_cast(u32x4, h); lo = __builtin_bit_cast(u32x4, l);
    } else {
        b8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (__bf16)v[j];
        hi = __builtin_bit_cast(u32x4, h); lo = hi;
    }
}
// LeakyReLU of two result tiles (8 values per lane) + their hi / lo f16 fragments, the encoder's conv -> Linear hand-over, written
// with the packed / mixed-precision instructions the compiler does not pick for the scalar formulation: per PAIR of values one
// v_pk_mul_f32 (0.01 x), two v_max_f32, v_cvt_pk_f16_f32 (hi), two v_fma_mix_f32 (lo = x - hi: the f16 halves are read in
// place, no conversion back) and v_cvt_pk_f16_f32 (lo): 28 vector instructions per 8 values instead of 39.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void leaky_split8(const f32x4& t0, const f32x4& t1, u32x4& hi, u32x4& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2 c = {0.01f, 0.01f};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2 x = p < 2 ? f32x2{t0[2 * p], t0[2 * p + 1]} : f32x2{t1[2 * p - 4], t1[2 * p - 3]};
        // The FIRST instruction that touches x is compiler-visible: t0 / t1 are MFMA results, and gfx950 does not interlock an MFMA's
        // result write against a vector instruction that reads the register -- the compiler's hazard recogniser inserts the wait
        // states, but it does not look into inline asm.  With `v_pk_mul_f32` as asm the 5-batch-tile 15 x 15 instantiation read the last
        // conv accumulator of batch tile 0 two instructions behind its MFMA (round 4: every row 0..15 of every workgroup 1e-3 off at
        // more than 32 768 rows; round 3's 16-bit-plane variant failed the same way, intermittently).  Everything below depends on y.
        const f32x2 y = x * c;
        float m0, m1, l0, l1;
        asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(x.x), "v"(y.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(x.y), "v"(y.y));
        uint32_t h, l;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(m0), "v"(m1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(m0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(m1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
        hi[p] = h; lo[p] = l;
    }
#else
    (void)t0; (void)t1; hi = u32x4{0u, 0u, 0u, 0u}; lo = hi;
#endif
}

// The same hand-over for SIX values that are vector-ALU results (the class-LUT encoder's channel sums; never an MFMA accumulator:
// see the hazard note above) + two zeros: 21 vector instructions instead of ~39.
__device__ __forceinline__ void leaky_split6(const f32x2 (&x3)[3], u32x4& hi, u32x4& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2 c = {0.01f, 0.01f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const f32x2 x = x3[p];
        const f32x2 y = x * c;
        float m0, m1, l0, l1;
        asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(x.x), "v"(y.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(x.y), "v"(y.y));
        uint32_t h, l;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(m0), "v"(m1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(m0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(m1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
        hi[p] = h; lo[p] = l;
    }
    hi[3] = 0u; lo[3] = 0u;
#else
    (void)x3; hi = u32x4{0u, 0u, 0u, 0u}; lo = hi;
#endif
}

// err: the device's numeric-status word (ssd_numeric_status); a scaled term outside f16's range (or NaN) raises ERR_F16_RANGE
template <int PREC> __device__ __forceinline__ void store_term(uint8_t* dst, float v, size_t term_stride, int32_t* err) {
    if constexpr (PREC == 2) {
        if (!(fabsf(v) <= F16_MAX) && err) atomicOr(err, ERR_F16_RANGE);
        const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
        *reinterpret_cast<_Float16*>(dst) = h; *reinterpret_cast<_Float16*>(dst + term_stride) = l;
    } else {
        *reinterpret_cast<__bf16*>(dst) = (__bf16)v;
    }
}

// Diagnostic build (-DSSD_STAMPS, tools/pstamps.py): per-wave s_memtime stamps of the kernel phases, [workgroup * waves + wave][16]
// u64, into a buffer registered with ssd_debug_set_policy_stamps.  In the product build the macros are empty: no stamp executes.
#ifdef SSD_STAMPS
static unsigned long long* g_policy_stamps = nullptr;
void set_policy_stamps(unsigned long long* buf) { g_policy_stamps = buf; }
#define PSTAMP_DECL unsigned long long* stamps;
#define PSTAMP_SET(k) (k).stamps = g_policy_stamps
#define PSTAMP_AT(i, v) do { if (a.stamps && lane == 0) a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave) * 16 + (i)] = (v); } while (0)
#define PSTAMP(i) PSTAMP_AT(i, __builtin_amdgcn_s_memtime())
#define PSTAMP_REAL(i) PSTAMP_AT(i, __builtin_amdgcn_s_memrealtime())      // chip-wide 100 MHz clock: comparable across XCDs
// stamp AFTER every outstanding memory operation of the wave has completed (prices a load phase; perturbs what follows)
#define PSTAMP_DRAINED(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); PSTAMP(i); } while (0)
#else
#define PSTAMP_DECL
#define PSTAMP_SET(k)
#define PSTAMP(i)
#define PSTAMP_REAL(i)
#define PSTAMP_DRAINED(i)
#endif

// exact power-of-two scales (PREC 2; PREC 1 needs none: bf16 has the f32 exponent range)
constexpr float HEAD_WSCALE = 64.f, HEAD_XSCALE = 16.f;                  // head weights / activations
constexpr float ENC_CSCALE = 256.f, ENC_LSCALE = 256.f;                  // conv weights (= conv activations' scale) / Linear weights

// ===========================================================================================================================
// heads
// ===========================================================================================================================
constexpr int HF_WI = 8, HF_WH = 32, HF_FC2 = 56, HF_TOT = SSD_POLICY_HEAD_FRAGS;
constexpr int HT_B1 = 0, HT_BI = 64, HT_BH = 256, HT_B2 = 448, HT_W2O = 464, HT_TOT = SSD_POLICY_HEAD_TAIL_FLOATS;
constexpr int SCRATCH = 16 * 16;               // per wave: fc2 output tile [row 16][out 16]
// The image is laid out in the order the head consumes it (include/ssd_hip.h), in 1 KiB pieces, 8 pieces to a CHUNK: per K-STEP c of
// the chain (fc1: c = s; GRU input side: 2 + 2 g + s; hidden side: 8 + 2 g + s) its [term][output tile] fragments (PREC 2: one chunk
// per K-step; PREC 1: two K-steps per chunk), then the RESIDENT chunk: fc2's [term][s] fragments, the f32 tail (3 pieces), padding.
// The image is brought in by a dedicated LOADER wave (the last wave of the workgroup; it owns no tile): vmcnt retires a wave's
// loads, stores and LDS-DMA pieces in issue order, so a wave that requested the image could not have its tile's inputs before all of
// it -- with the loader the compute waves' own memory traffic is untouched and hipcc's wait counts stay exact.  The loader requests
// chunk after chunk with 6 chunks (48 KiB) in flight, whatever the compute waves are doing -- the ingest (~11 bytes per cycle and CU
// from the Infinity Cache) runs under their input phase -- and publishes its progress in an LDS word; a compute wave looks at that word
// before the first read of a chunk and only waits when the stream is behind it.  No workgroup barrier anywhere: the waves of a
// workgroup share nothing but the image.
constexpr int HEAD_STEPS = 14;
template <int PREC> constexpr int stream_chunks() { return HEAD_STEPS * 4 * PREC / 8; }                       // 14 / 7
template <int PREC> constexpr int step_piece(int c, int t, int ot) { return 4 * PREC * c + 4 * t + ot; }
template <int PREC> constexpr int res_piece() { return stream_chunks<PREC>() * 8; }                         // first piece of the resident chunk
template <int PREC> constexpr int fc2_piece(int t, int s2) { return res_piece<PREC>() + 2 * t + s2; }
template <int PREC> constexpr int tail_piece() { return res_piece<PREC>() + 2 * PREC; }
static_assert(res_piece<2>() + 8 == SSD_POLICY_IMAGE_PIECES(2) && res_piece<1>() + 8 == SSD_POLICY_IMAGE_PIECES(1), "image = stream chunks + resident chunk");
static_assert(tail_piece<2>() + SSD_POLICY_TAIL_PIECES <= SSD_POLICY_IMAGE_PIECES(2) && tail_piece<1>() + SSD_POLICY_TAIL_PIECES <= SSD_POLICY_IMAGE_PIECES(1), "tail fits");
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// One LDS-DMA piece: 64 lanes x 16 bytes from each lane's global address into 1 KiB of LDS at the wave-uniform byte address lds_dst
// (global_load_lds_dwordx4: no VGPR destination; M0 carries the LDS address and is restored).  hipcc does not count it: completion is
// waited for with wait_vm<N> below, then a workgroup barrier, then the ds_reads.
__device__ __forceinline__ void dma_piece(const uint8_t* gsrc_lane, uint32_t lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
#else
    (void)gsrc_lane; (void)lds_dst;
#endif
}
template <int N> __device__ __forceinline__ void wait_vm() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
#endif
}
// The progress word is accessed as an LDS word (ds_write_b32 / ds_read_b32).  Through a generic pointer the same accesses are FLAT
// instructions, which count in vmcnt as well: the loader's volatile store was followed by s_waitcnt vmcnt(0) -- every publish waited for
// ALL chunks in flight, so behind the first burst the image came in one chunk at a time -- and a compute wave's poll waited for its
// own outstanding global loads.
using lds_u32 = __attribute__((address_space(3))) uint32_t;
constexpr int STREAM_AHEAD = 6;                // chunks in flight behind the one being waited for (x 8 pieces < the 64 vmcnt can count)
constexpr int head_lds_bytes(int waves, int prec) { return SSD_POLICY_IMAGE_BYTES(prec) + waves * SCRATCH * 4 + 16; }
// the loader wave's whole program: LDS image = the global image, piece for piece (chunk j at byte 8192 j, the resident chunk last)
template <int PREC>
__device__ __forceinline__ void stream_image(const uint8_t* src_lane, uint32_t lds_dst, volatile lds_u32* landed) {
    constexpr int NK = stream_chunks<PREC>(), NC = NK + 1;            // + the resident chunk, requested FIRST (fc1's biases are in it)
    auto issue = [&](int k) {                                          // k-th request: resident chunk, then stream chunks 0 .. NK - 1
        const int j = k == 0 ? NK : k - 1;
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) dma_piece(src_lane + (size_t)j * 8192 + pc * 1024, lds_dst + (uint32_t)j * 8192u + (uint32_t)pc * 1024u);
    };
#pragma unroll
    for (int k = 0; k < cmin(STREAM_AHEAD + 1, NC); ++k) issue(k);
#pragma unroll
    for (int k = 0; k < NC; ++k) {                                     // request k has landed (everything older too) -> publish, request the next
        if (k + STREAM_AHEAD < NC) wait_vm<8 * STREAM_AHEAD>();
        else if (k == NC - 1) wait_vm<0>();
        else if (k == NC - 2) wait_vm<8>();
        else if (k == NC - 3) wait_vm<16>();
        else if (k == NC - 4) wait_vm<24>();
        else if (k == NC - 5) wait_vm<32>();
        else wait_vm<40>();
        *landed = (uint32_t)k;                                         // = stream chunks 0 .. k - 1 (and the resident chunk) are readable
        if (k + STREAM_AHEAD + 1 < NC) issue(k + STREAM_AHEAD + 1);
    }
}
// a compute wave, before the first read of K-step C's chunk j: the loader must have published `landed` > j (chunks 0 .. j readable;
// seen caches the last value read, so a wave that runs behind the stream reads the word once)
template <int PREC, int C> __device__ __forceinline__ void chunk_ready(volatile const lds_u32* landed, uint32_t& seen) {
    constexpr int first_piece = step_piece<PREC>(C, 0, 0), j = first_piece >> 3;
    if constexpr ((first_piece & 7) == 0) {
        while (seen <= (uint32_t)j) {
            seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)*landed);
            if (seen <= (uint32_t)j) __builtin_amdgcn_s_sleep(1);
        }
        // compiler barrier: the image's (non-volatile) ds_reads that follow must not be hoisted above the poll -- `volatile` orders the
        // poll only against other volatile accesses.  No instruction is emitted (LDS returns in order, so the hardware needs no wait)
        asm volatile("" ::: "memory");
    }
}

// Kernel arguments.  HeadK is what the main path reads; HeadCold holds the ~20 pointers only the epilogue touches (results, filing
// into the episode storage, runner state).  hipcc loads every by-value kernel argument into SGPRs at kernel entry -- 100+ SGPRs
// here, most of them then spilled to VGPR lanes for the whole kernel -- so the cold ones are fetched from the kernarg segment
// where they are used (cold_ptr below: one s_load_dwordx2 each).
struct HeadK {
    int N, n, A, inp, bpa, slots, feat_bands;
    float pos_scale;
    uint32_t seed, env_id_base;
    float* inputs;
    float* h;
    const uint8_t* weights;
    const uint8_t* avail;
    const float* eps;
    const int64_t* step;
    const int64_t* prev_actions;
    const float* prev_reward;
    const int64_t* prev_inc;
    const float *pos, *orient;
    const int64_t* actions;
    const float *pos_pre, *orient_pre, *reward, *clean, *den;
    const int64_t* t_index;
    const float *feat_part, *lin_b;
    const float* ep_ret;           // inc head: the episode return so far (read with the tile's inputs, written back in the epilogue)
    const uint8_t* term;           // inc head: the env step's terminated flags
    const uint8_t* recv;           // env head, nullable: receiver-major incentive bytes [n, N, 16] (instead of prev_inc)
    uint32_t avail_bits;           // env head: bit 31 set = the availability mask itself (no loads from `avail`)
    PSTAMP_DECL
};
struct HeadCold {
    // block A (64 bytes): what both epilogues write first, then the env head's results and by-products
    int64_t* out_actions; float* q_out; int32_t* out_actions_i32; int64_t* p_act; int64_t* d_actions; float* d_onehot; float *pos_copy, *orient_copy;
    // block B (64 bytes): the inc head's filing
    int64_t *p_inc, *d_actions_inc; float *d_reward, *p_rew, *ep_ret, *d_clean, *d_den; uint8_t* d_term;
    // block C (16 bytes): the env head's pose filing
    float *d_pos, *d_orient;
    int32_t* numeric_err;          // the device's numeric-status word (ERR_F16_RANGE)
    int64_t* next_t;
    int64_t *next_step, *t_copy, *step_copy;   // inc head: *next_step = *step + 1; env head: *t_copy = *t_index, *step_copy = *step
    uint8_t* recv_out;             // inc head, nullable: receiver-major copy of the incentive actions (byte `agent` of [j, N, 16])
    uint64_t tail_layout;          // env head: first tail column of each _build_inputs block as 6 signed bytes (TAIL_ABSENT = not present):
                                   // last action | agent id | sign(r) | sign(received incentives) | 1 - distances | pos
};
static_assert(offsetof(HeadCold, p_inc) == 64 && offsetof(HeadCold, d_pos) == 128, "cold argument blocks");
constexpr int TAIL_ABSENT = -64;
constexpr int HEAD_COLD_OFFSET = (int)((sizeof(HeadK) + alignof(HeadCold) - 1) / alignof(HeadCold) * alignof(HeadCold));   // second kernel argument
// The cold pointers are GLOBAL pointers by type: assembled from scalar loads the compiler knows nothing about their address space
// and a plain `T*` becomes a FLAT access -- which counts in lgkmcnt as well as in vmcnt, so that the next wait for an LDS read or
// a scalar load (every K-step has one) would also wait for every by-product store in flight.
// (cold_base: the offset of the launch's HeadCold argument in its kernarg segment = the kernel's leading scalars + HEAD_COLD_OFFSET)
template <typename T>
__device__ __forceinline__ T SSD_GLOBAL* cold_ptr(int cold_base, int field_offset) {
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    uint64_t v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ka), "s"(cold_base + field_offset) : "memory");
    return (T SSD_GLOBAL*)v;
}
// Several cold pointers at once: ONE scalar-load burst and ONE wait instead of a load + wait per pointer (~150 cycles each on a wave
// that is alone on its SIMD).  cold_blocks: block A [+ block B] [+ block C] of HeadCold.
typedef uint32_t u32x16s __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
struct ColdA { int64_t SSD_GLOBAL* out_actions; float SSD_GLOBAL* q_out; int32_t SSD_GLOBAL* out_actions_i32; int64_t SSD_GLOBAL* p_act; int64_t SSD_GLOBAL* d_actions;
               float SSD_GLOBAL* d_onehot; float SSD_GLOBAL *pos_copy, *orient_copy; };
struct ColdB { int64_t SSD_GLOBAL *p_inc, *d_actions_inc; float SSD_GLOBAL *d_reward, *p_rew, *ep_ret, *d_clean, *d_den; uint8_t SSD_GLOBAL* d_term; };
struct ColdC { float SSD_GLOBAL *d_pos, *d_orient; };
__device__ __forceinline__ void cold_blocks(int cold_base, ColdA& A, ColdB* B, ColdC* C) {
#if defined(__HIP_DEVICE_COMPILE__)
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    u32x16s a, b = {};
    u32x4s c = {};
    if (B && C) asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx4 %2, %3, %6\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(ka), "s"(cold_base), "s"(cold_base + 64), "s"(cold_base + 128) : "memory");
    else if (B) asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(ka), "s"(cold_base), "s"(cold_base + 64) : "memory");
    else if (C) asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(c) : "s"(ka), "s"(cold_base), "s"(cold_base + 128) : "memory");
    else asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(ka), "s"(cold_base) : "memory");
    A = __builtin_bit_cast(ColdA, a);
    if (B) *B = __builtin_bit_cast(ColdB, b);
    if (C) *C = __builtin_bit_cast(ColdC, c);
#else
    (void)A; (void)B; (void)C;
#endif
}
#define COLD(T, field) cold_ptr<T>(COLD_BASE, (int)offsetof(HeadCold, field))
#define COLD_U64(field) ((uint64_t)cold_ptr<void>(COLD_BASE, (int)offsetof(HeadCold, field)))

// One K-step (c) of a 4-output-tile product as two halves, so that a SEQUENCE of products can request the next step's A fragments
// from LDS before the current step's MFMAs issue (a wave that is alone on its SIMD has nothing else to hide the ds_read latency):
// frag4_load fetches the 4 (PREC 2: 8) fragments of K-step c, frag4_mma adds their products.
template <int PREC>
struct Frag4 { u32x4 h[4], l[4]; };
template <int PREC>
__device__ __forceinline__ void frag4_load(const uint8_t* img, int c, int lane, Frag4<PREC>& f) {
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        const uint8_t* p = img + (size_t)step_piece<PREC>(c, 0, ot) * 1024 + lane * 16;
        f.h[ot] = *reinterpret_cast<const u32x4*>(p);
        if (PREC == 2) f.l[ot] = *reinterpret_cast<const u32x4*>(p + 4 * 1024);
    }
}
template <int PREC>
__device__ __forceinline__ void frag4_mma(const Frag4<PREC>& f, u32x4 bh, u32x4 bl, f32x4* acc) {
    if (PREC == 2) {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) acc[ot] = mma<PREC>(f.l[ot], bh, acc[ot]);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) acc[ot] = mma<PREC>(f.h[ot], bl, acc[ot]);
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mma<PREC>(f.h[ot], bh, acc[ot]);
}

// the B operand of a product from 4 x 4 features per lane (x[ct][r] = feature 16 ct + 4 q + r), scaled
template <int PREC>
__device__ __forceinline__ void operand(const f32x4 (&x)[4], float scale, u32x4 (&bh)[2], u32x4 (&bl)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[2 * s + (j >> 2)][j & 3] * scale;
        split8<PREC>(v, bh[s], bl[s]);
    }
}

// running max |x| over result tiles (one v_max3_f32 with |.| source modifiers per two values): the range guard of the activations that
// are about to be scaled and split (the fc1 operand and fc1's output; hidden states are < 1 by construction)
__device__ __forceinline__ float amax2(float acc, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
#else
    return fmaxf(acc, fmaxf(fabsf(a), fabsf(b)));
#endif
}
template <int NT>
__device__ __forceinline__ float amax_tiles(float acc, const f32x4 (&x)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc = amax2(acc, x[t][0], x[t][1]); acc = amax2(acc, x[t][2], x[t][3]); }
    return acc;
}

// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (1 ulp each)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fminf(fabsf(x), 15.f);                            // 1 - 2 / (e^{2|x|} + 1), saturated
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * ax) + 1.f);
    return copysignf(t, x);
}

// Everything a 16-row tile needs from global memory, requested in ONE batch of independent loads: the first tile's batch is
// issued before the weight image is staged (its latency hides under the staging), later ones right after the previous tile.
template <int INC>
struct TileIn {
    f32x4 x[4];                    // env: x[0..1] = encoder features (final, or lin_b + the band sums); inc: the 64 stored inputs
    f32x4 hp[4];                   // previous hidden state
    int pa, act;                   // env: last action; inc: the env action just taken
    int32_t inc[3];                // env: the incentive action giver g = q + 4 k sent this agent at the previous step (0 / 1 / 2), k = 0 .. 2:
                                   // the four lane quarters of a row share the givers between them (summed over q in prepare)
    uint32_t rv;                   // env, receiver-major bytes given: givers 4 q .. 4 q + 3 as one dword
    float pr, p0, p1, o0, o1;      // env: last reward, pose
    int aj[3];                     // inc epilogue items (row, j) = lane + 64 k: action of j and its 7 features
    float f[3][7];
    float own[3], ep;              // inc, lanes < 16 (row = lane): this agent's reward / clean_num / apple_den of the step, its return so far
    int term;                      // inc, agent 0: the env's terminated flag
};

// Addresses are 32-bit element offsets from the (scalar) base pointers -- every array here has far fewer than 2^31 elements (the ABI
// checks n_env * n_agents^2 * 64) -- so a load is `global_load v, v_offset, s[base]` instead of a chain of 64-bit multiplies per lane,
// and nothing in here branches: a branch the compiler makes out of a predicated load waits for EVERY outstanding load of the wave,
// the image's first chunks included (measured: 5 300 of the prologue's 12 000 cycles).
template <typename T>
__device__ __forceinline__ T ld32(const T* base, uint32_t idx) {
    // (a GLOBAL access by type: the looped kernels rebuild HeadK from dwords of the kernarg segment, which leaves the compiler with
    // generic pointers -- and FLAT loads, which count in lgkmcnt as well: every wait for an LDS read would wait for them too)
    typedef const T SSD_GLOBAL gT;
    return *(gT*)((const uint8_t SSD_GLOBAL*)(const void SSD_GLOBAL*)base + (size_t)(idx * (uint32_t)sizeof(T)));
}
__device__ __forceinline__ void stg128(void* base, size_t byte_off, f32x4 v) {
    typedef f32x4 SSD_GLOBAL gv;
    *(gv*)((uint8_t SSD_GLOBAL*)(void SSD_GLOBAL*)base + byte_off) = v;
}
__device__ __forceinline__ f32x4 ldg128(const void* base, size_t byte_off) {
    typedef const f32x4 SSD_GLOBAL gv;
    return *(gv*)((const uint8_t SSD_GLOBAL*)(const void SSD_GLOBAL*)base + byte_off);
}
// The tile's largest loads -- the agent-major state row and (unless the encoder left band partials) the input row: they need
// nothing but the kernel's leading scalar arguments (k_head), so the kernels without a back edge request them before anything else.
// SMALL (the standalone env head): also the tail's scalars -- previous action / reward, the receiver-major incentive bytes, the
// position -- whose pointers are leading arguments of k_head too: every load the input assembly waits for is then in flight
// before the first trip to the kernarg segment returns.
template <int INC, bool SMALL = false>
__device__ __forceinline__ void load_tile_rows(const HeadK& a, int tile, int agent, int lane, TileIn<INC>& in) {
    const int m = lane & 15, q = lane >> 4, N = a.N;
    const int b = tile * 16 + m, bc = b < N ? b : N - 1;
    const uint32_t arow = (uint32_t)agent * (uint32_t)N + (uint32_t)bc;
    const uint32_t ro = arow * 64u + 4u * (uint32_t)q;                 // this lane's first float of the agent-major row
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) in.hp[ct] = ldg128(a.h, (size_t)((ro + 16u * ct) * 4u));
#pragma unroll
    for (int ct = 0; ct < (INC ? 4 : 2); ++ct) in.x[ct] = ldg128(a.inputs, (size_t)((ro + 16u * ct) * 4u));
    if constexpr (SMALL && !INC) {
        const uint32_t er = (uint32_t)bc * (uint32_t)a.n + (uint32_t)agent;   // env-major row
        in.pa = (int)ld32(reinterpret_cast<const int32_t*>(a.prev_actions), 2u * er);
        in.pr = ld32(a.prev_reward, er);
        in.rv = a.recv ? ld32(reinterpret_cast<const uint32_t*>(a.recv), arow * 4u + (uint32_t)q) : 0u;
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const f32x2v pp = ld32(reinterpret_cast<const f32x2v*>(a.pos), er);
        in.p0 = pp.x; in.p1 = pp.y;
    }
}
// ROWS_DONE: load_tile_rows already requested the state row and (feat_part == nullptr or INC) the input row
template <int INC, bool ROWS_DONE = false, bool SMALL_DONE = false>
__device__ __forceinline__ void load_tile(const HeadK& a, int tile, int agent, int lane, TileIn<INC>& in) {
    const int m = lane & 15, q = lane >> 4, N = a.N, n = a.n;
    const int b = tile * 16 + m, bc = b < N ? b : N - 1;
    const uint32_t arow = (uint32_t)agent * (uint32_t)N + (uint32_t)bc;
    const uint32_t ro = arow * 64u + 4u * (uint32_t)q;                 // this lane's first float of the agent-major row
    if (!ROWS_DONE) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) in.hp[ct] = ldg128(a.h, (size_t)((ro + 16u * ct) * 4u));
    }
    if (!INC) {
        if (a.feat_part) {          // the encoder left per-band partial sums: lin_b + sum over the bands, band order (wave-uniform branch)
            const uint32_t rows = (uint32_t)n * (uint32_t)N;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 part[6];
#pragma unroll
                for (int bd = 0; bd < 6; ++bd) {   // branch-free: bands past the last re-read the last one and are masked out
                    const int bdc = bd < a.feat_bands ? bd : a.feat_bands - 1;
                    part[bd] = ldg128(a.feat_part, (size_t)((((uint32_t)bdc * rows + arow) * 32u + 16u * ct + 4u * q) * 4u));
                }
                f32x4 s = ldg128(a.lin_b, (size_t)((16 * ct + 4 * q) * 4));
#pragma unroll
                for (int bd = 0; bd < 6; ++bd) { const float on = bd < a.feat_bands ? 1.f : 0.f; s += part[bd] * on; }
                in.x[ct] = s;
            }
        } else if (!ROWS_DONE) {
            in.x[0] = ldg128(a.inputs, (size_t)(ro * 4u));
            in.x[1] = ldg128(a.inputs, (size_t)((ro + 16u) * 4u));
        }
        const uint32_t er = (uint32_t)bc * (uint32_t)n + (uint32_t)agent;     // env-major row
        if (!SMALL_DONE) {
            in.pa = (int)ld32(reinterpret_cast<const int32_t*>(a.prev_actions), 2u * er);   // low word of the int64 (little endian; -1 .. A - 1)
            in.pr = ld32(a.prev_reward, er);
        }
        // received incentives: prev_inc[bc, g, agent] for every giver g (values 0 / 1 / 2: the low word); givers past n re-read giver
        // n - 1 and count nothing
        if (a.recv) {                                                  // (wave-uniform) the row's 16 giver bytes: dword q for lane quarter q
            if (!SMALL_DONE) in.rv = ld32(reinterpret_cast<const uint32_t*>(a.recv), arow * 4u + (uint32_t)q);
            in.inc[0] = in.inc[1] = in.inc[2] = 0;
        } else {
            in.rv = 0u;
            const int32_t* pi = reinterpret_cast<const int32_t*>(a.prev_inc);
#pragma unroll
            for (int k = 0; k < 3; ++k) {                              // (counted in prepare: nothing here waits for a load)
                const int g = q + 4 * k;
                const uint32_t gc = (uint32_t)(g < n ? g : n - 1);
                in.inc[k] = (k == 0 || n > 4 * k) ? ld32(pi, 2u * (((uint32_t)bc * (uint32_t)n + gc) * (uint32_t)n + (uint32_t)agent)) : 0;   // wave-uniform predicate
            }
        }
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        if (!SMALL_DONE) {
            const f32x2v pp = ld32(reinterpret_cast<const f32x2v*>(a.pos), er);      // one 8-byte load per pair
            in.p0 = pp.x; in.p1 = pp.y;
        }
        const float* orient = a.orient ? a.orient : a.pos;             // no orientation given: read something valid; file_inputs writes zeros
        const f32x2v oo = ld32(reinterpret_cast<const f32x2v*>(orient), er);
        in.o0 = oo.x; in.o1 = oo.y;                                    // (raw: an arithmetic use here would be a wait for every load of the tile)
    } else {
        if (!ROWS_DONE) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) in.x[ct] = ldg128(a.inputs, (size_t)((ro + 16u * ct) * 4u));
        }
        in.act = (int)ld32(reinterpret_cast<const int32_t*>(a.actions), 2u * ((uint32_t)bc * (uint32_t)n + (uint32_t)agent));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int it = lane + 64 * k, row = it / n, j = it - row * n, bb = tile * 16 + row;
            // other_j = [one-hot(a_j), pos_j / scale, orient_j, r_j, clean_j, apple_den_j] (homophily_agent.py:194-201); items past the
            // tile (or the batch) read the batch's last row and are never used
            const bool live = it < 16 * n && bb < N;
            const uint32_t ej = live ? (uint32_t)bb * (uint32_t)n + (uint32_t)j : (uint32_t)N * (uint32_t)n - 1u;
            in.aj[k] = (int)ld32(reinterpret_cast<const int32_t*>(a.actions), 2u * ej);
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v pj = ld32(reinterpret_cast<const f32x2v*>(a.pos_pre), ej), oj = ld32(reinterpret_cast<const f32x2v*>(a.orient_pre), ej);
            in.f[k][0] = pj.x; in.f[k][1] = pj.y; in.f[k][2] = oj.x; in.f[k][3] = oj.y;
            in.f[k][4] = ld32(a.reward, ej); in.f[k][5] = ld32(a.clean, ej); in.f[k][6] = ld32(a.den, ej);
        }
        {   // once per (env, agent) -- lanes < 16, row = lane: what the epilogue files for the agent itself
            const int br = tile * 16 + (lane & 15), brc = br < N ? br : N - 1;
            const uint32_t ea = (uint32_t)brc * (uint32_t)n + (uint32_t)agent;
            in.own[0] = ld32(a.reward, ea); in.own[1] = ld32(a.clean, ea); in.own[2] = ld32(a.den, ea);
            in.ep = ld32(a.ep_ret ? a.ep_ret : a.reward, ea);
            in.term = (int)ld32(a.term ? a.term : reinterpret_cast<const uint8_t*>(a.reward), (uint32_t)brc);
        }
    }
}

// AT: the env's action count (9 Cleanup, 8 Harvest) at compile time: the one-hot / dueling loops unroll.  GEN (env head): the
// tail blocks sit where HeadCold::tail_layout says (any _build_inputs flag set that fits); GEN = 0 is the shipped layout with
// compile-time columns -- the register allocation of the tuned kernel is left as it was.
// WAVES: compute waves per workgroup = 16-row tiles per workgroup (the host sizes the grid: bpa = ceil(tiles per agent / WAVES)); the
// workgroup has WAVES + 1 waves, the last one is the loader.
// LOOP: a wave walks tiles tile, tile + bpa * WAVES, ... (grids larger than the chip: Cleanup-10 x 8192 has 512 tiles per agent); with
// LOOP = false the host guarantees at most one tile per wave and the kernel has no back edge (see run_tile).
// The kernel arguments once more, from the kernarg segment (HeadK is the FIRST argument of k_head and of k_inc_encode): the address is
// made opaque to the compiler, so the scalar loads are issued where the call stands and their results are not values that were live
// since kernel entry.  The looped kernels call this at the top of every pass over a tile (see head_body).
__device__ __forceinline__ void refetch_head_args(HeadK& out, int k_off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const uint32_t k_u32;
    uint64_t p = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    k_u32* src = (k_u32*)(p + (uint64_t)k_off);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(HeadK) / 4); ++i) dst[i] = src[i];
#else
    (void)out;
#endif
}
static_assert(sizeof(HeadK) % 4 == 0, "HeadK is copied dword by dword");

// KOFF: the offset of the HeadK argument in the launch's kernarg segment (behind the kernel's leading scalar arguments)
template <int INC, int PREC, int AT, int GEN, int WAVES, bool LOOP = false, int KOFF = 0, bool PRE_SMALL = false>
__device__ __forceinline__ void head_body(const HeadK& a_entry, uint8_t* lds_raw, const int block) {
    constexpr int COLD_BASE = KOFF + HEAD_COLD_OFFSET;
    constexpr int IMAGE_BYTES = SSD_POLICY_IMAGE_BYTES(PREC);
    constexpr float XS = PREC == 2 ? HEAD_XSCALE : 1.f, INV = PREC == 2 ? 1.f / (HEAD_WSCALE * HEAD_XSCALE) : 1.f;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int A = AT;
    uint32_t seen = 0;                                                 // last value read from the loader's progress word (kept across passes)
    bool first = true;
    // LDS: the image (as in global memory) | per-wave scratch | the loader's progress word
    volatile lds_u32* landed = (volatile lds_u32*)(lds_raw + IMAGE_BYTES + WAVES * SCRATCH * 4);      // generic -> LDS address space
    TileIn<INC> in_rows;                                               // !LOOP: the tile's row loads, requested in the prologue
    {
        const HeadK& a = a_entry;
        const int lane = tid & 63;
        (void)lane;
        PSTAMP(0);
        PSTAMP_REAL(14);
        if (!LOOP && wave < WAVES) {   // (see load_tile_rows; the loader wave has no tile)
            const int bia0 = block - (block / a.bpa) * a.bpa, tile00 = wave * a.bpa + bia0;
            if (tile00 < ((a.N + 15) >> 4)) load_tile_rows<INC, PRE_SMALL>(a, tile00, block / a.bpa, lane, in_rows);
        }
        // Every kernel argument the input phase reads, requested in ONE batch with the first one (by-value arguments are fetched from
        // the kernarg segment where they are first used: three dependent scalar-memory trips, ~ 1 K cycles each while the segment
        // is cold, stand between the wave's start and its first tile load otherwise).
        if (!LOOP) {
            if (!INC) asm volatile("" :: "s"(a.N), "s"(a.n), "s"(a.bpa), "s"(a.h), "s"(a.inputs), "s"(a.weights), "s"(a.eps), "s"(a.step), "s"(a.t_index),
                                         "s"(a.prev_actions), "s"(a.prev_reward), "s"(a.recv), "s"(a.prev_inc), "s"(a.pos), "s"(a.orient), "s"(a.feat_part),
                                         "s"(a.lin_b), "s"(a.feat_bands), "s"(a.avail_bits), "s"(a.pos_scale));
            else asm volatile("" :: "s"(a.N), "s"(a.n), "s"(a.bpa), "s"(a.h), "s"(a.inputs), "s"(a.weights), "s"(a.eps), "s"(a.step), "s"(a.t_index),
                                    "s"(a.actions), "s"(a.pos_pre), "s"(a.orient_pre), "s"(a.reward), "s"(a.clean), "s"(a.den), "s"(a.ep_ret), "s"(a.term),
                                    "s"(a.inp), "s"(a.pos_scale));
        }
        if (wave == WAVES) *landed = 0u;                               // LDS holds garbage at launch: the word is valid behind this barrier
        __builtin_amdgcn_s_barrier();                                  // (the only one: every wave is still at its first instructions)
        if (wave == WAVES) {                                           // the loader wave: nothing but the image stream
            const int agent = block / a.bpa;
            stream_image<PREC>(a.weights + (size_t)agent * IMAGE_BYTES + lane * 16, (uint32_t)(uintptr_t)lds_raw, landed);
            return;
        }
    }
    // One PASS = one 16-row tile from its loads to its stores.  `a` and `lane` are parameters so that the looped kernels can hand every
    // pass freshly fetched arguments and an opaque lane id: nothing of a pass is then invariant across the back edge, the compiler
    // hoists nothing out of the loop and keeps no argument alive through the tile chain -- the pass has the register footprint of the
    // kernel without a loop.  tile >= tiles: a wave without a tile (it only hands the counters over).
    auto pass = [&](const HeadK& a, const int lane, const int tile) {
    const int agent = block / a.bpa;
    const int m = lane & 15, q = lane >> 4;
    const int N = a.N, n = a.n;
    const int tiles = (N + 15) >> 4;
    TileIn<INC> in;
    if constexpr (!LOOP) in = in_rows;
    if (!INC) PSTAMP_DRAINED(8);                                       // (diagnostic builds: kernel arguments fetched)
    if (!INC) PSTAMP(9);
    // the device-side counters first: scalar loads through pointers, in flight under the tile's loads instead of behind them
    float eps = *(const float SSD_GLOBAL*)a.eps;
    int64_t step64 = *(const int64_t SSD_GLOBAL*)a.step;
    long slot_t = a.t_index ? (long)*(const int64_t SSD_GLOBAL*)a.t_index : 0;
    if (tile < tiles) {                                                // queued behind the first chunks
        if constexpr (LOOP) load_tile<INC, false>(a, tile, agent, lane, in);
        else if (!INC && a.feat_part) { load_tile<INC, false, PRE_SMALL>(a, tile, agent, lane, in); }     // (band partials instead of the input row: Harvest)
        else load_tile<INC, true, PRE_SMALL>(a, tile, agent, lane, in);
    }
    if (!INC) PSTAMP(10);
    uint32_t step = 0;                                                 // (both assigned once the loads above are pinned, see below)
    bool file = false;
    const uint32_t avail_bits = INC ? 0xFFFFFFFFu : ((a.avail_bits >> 31) ? (a.avail_bits & 0x7FFFFFFFu) : avail_to_bits(a.avail, A));
    if (!INC) PSTAMP_DRAINED(11);
    u32x4 bh[2], bl[2];                                                // the fc1 operand of the current tile
    // range guard of the activations that get scaled by XS and split (PREC 2): checked where they are produced (no state carried
    // through the tile chain -- the env head has no register to spare); the branch is never taken on a healthy network
    auto range_check = [&](float amax) {
        if (!(amax * XS <= F16_MAX)) {
            auto nerr = COLD(int32_t, numeric_err);
            if (nerr) atomicOr((int32_t*)nerr, ERR_F16_RANGE);
        }
    };
    // Everything of a tile that only needs its inputs -- the input tail (controller :137-184) and the split of the fc1 operand -- is
    // done as soon as the loads land, while the rest of the weight image is in flight.  It issues NO store: vmcnt retires loads, stores
    // and LDS-DMA pieces in issue order, so a store issued behind the image's pieces could not complete before all of them -- and a
    // counted wait for the next chunk would wait for the whole image.  The by-product stores (file_inputs) follow the last K-step.
    f32x4 xk[4];                                                       // the tile's input row, kept for file_inputs
    auto prepare = [&](int tl) {
        const int b = tl * 16 + m;
        const bool valid = b < N;
        const int bc = valid ? b : N - 1;
        f32x4 (&x)[4] = xk;
        if (!INC) {
            x[0] = in.x[0]; x[1] = in.x[1];
            if (a.feat_part) {      // finish the encoder: LeakyReLU of (lin_b + band sums); the inc head reads them from `inputs`
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[ct][r] = leaky(x[ct][r]);
                }
            }
            // tail columns (controller :137-184, each block present iff its flag is set): one-hot(last action) | one-hot(agent id)
            // | sign(r) | sign(received incentives) | 1 - distance to every agent / scale | pos / scale; absent blocks sit at a
            // negative column and never match
            int o_act = 0, o_id = A, o_r = A + n, o_i = A + n + 1, o_dist = TAIL_ABSENT, o_pos = A + n + 2;
            if (GEN) {
                const uint64_t lay = COLD_U64(tail_layout);
                o_act = (int8_t)lay; o_id = (int8_t)(lay >> 8); o_r = (int8_t)(lay >> 16); o_i = (int8_t)(lay >> 24);
                o_dist = (int8_t)(lay >> 32); o_pos = (int8_t)(lay >> 40);
            }
            const float px = in.p0 / a.pos_scale, py = in.p1 / a.pos_scale;
            int recv = 0;                                              // #rewards - #punishments received (controller :150-157)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int g = q + 4 * k;
                const int on = (g < n && g != agent) ? 1 : 0;          // inc_mask_actions: no self incentive
                recv += on * ((in.inc[k] == 1) - (in.inc[k] == 2));
            }
            // receiver-major bytes: values 0 / 1 / 2, the receiver's own byte and the bytes of givers >= n are 0 (the inc head never
            // writes them), so the count is bit 0 of every byte minus bit 1 of every byte
            recv += __builtin_popcount(in.rv & 0x01010101u) - __builtin_popcount(in.rv & 0x02020202u);
            recv += __shfl_xor(recv, 16);                              // the row's four quarters hold different givers
            recv += __shfl_xor(recv, 32);
            const float sg_r = (float)((in.pr > 0.f) - (in.pr < 0.f)), sg_i = (float)((recv > 0) - (recv < 0));
            const int c_pa = o_act + in.pa, c_id = o_id + agent;       // pa in [-1, A): -1 (no previous step) lands left of the block
#pragma unroll
            for (int ct = 2; ct < 4; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * ct + 4 * q + r - 32;
                    float v = (j == c_pa || j == c_id) ? 1.f : 0.f;    // a hit is a one-hot column
                    v = j == o_r ? sg_r : v;
                    v = j == o_i ? sg_i : v;
                    v = j == o_pos ? px : v;
                    v = j == o_pos + 1 ? py : v;
                    x[ct][r] = v;
                }
            }
            if (GEN && o_dist >= 0) {      // obs_distance (wave-uniform; not in the shipped flag set): 1 - |pos_i - pos_g| / scale, g = 0..n-1
                for (int g = 0; g < n; ++g) {
                    const float dx = in.p0 - a.pos[((size_t)bc * n + g) * 2], dy = in.p1 - a.pos[((size_t)bc * n + g) * 2 + 1];
                    const float d = 1.f - sqrtf(dx * dx + dy * dy) / a.pos_scale;
#pragma unroll
                    for (int ct = 2; ct < 4; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[ct][r] = (16 * ct + 4 * q + r - 32 == o_dist + g) ? d : x[ct][r];
                }
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                x[ct] = in.x[ct];
                if (ct >= 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 16 * ct + 4 * q + r - a.inp;     // [inputs | one-hot(action)] (homophily_agent.py:181)
                        if (k >= 0 && k < A) x[ct][r] = in.act == k ? 1.f : 0.f;
                    }
                }
            }
        }
        operand<PREC>(x, XS, bh, bl);
        (void)valid; (void)bc;
    };
    // the env head's by-products of the input phase: the finished encoder features (31 x 31 windows) and the tail columns into the
    // agent's input row (the inc head reads the full row), the pose BEFORE the env step (inc head input, storage slot t)
    auto file_inputs = [&](int tl) {
        if (INC) return;
        const int b = tl * 16 + m;
        if (b >= N) return;
        float* in_row = a.inputs + ((size_t)agent * N + b) * 64;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (ct >= 2 || a.feat_part) stg128(in_row, (size_t)((16 * ct + 4 * q) * 4), xk[ct]);
        if (q == 0) {
            const size_t er = (size_t)b * n + agent;                   // env-major row
            ColdA ca; ColdC cc;
            cold_blocks(COLD_BASE, ca, nullptr, &cc);
            const float o0 = a.orient ? in.o0 : 0.f, o1 = a.orient ? in.o1 : 0.f;     // no orientation given: zeros
            if (ca.pos_copy) {
                ca.pos_copy[er * 2] = in.p0; ca.pos_copy[er * 2 + 1] = in.p1; ca.orient_copy[er * 2] = o0; ca.orient_copy[er * 2 + 1] = o1;
            }
            if (cc.d_pos && file) {
                const size_t sr = (((size_t)b * a.slots + slot_t) * n + agent) * 2;
                cc.d_pos[sr] = in.p0; cc.d_pos[sr + 1] = in.p1; cc.d_orient[sr] = o0; cc.d_orient[sr + 1] = o1;
            }
        }
    };
    // Device-side counters advance by ping-pong copies, never by a kernel incrementing a scalar it (or a workgroup of the same
    // launch) also reads: the inc head reads the copies and writes the masters' next values, the env head (pipelined rollout) reads
    // the masters and writes the copies.  (Stores: after the last K-step, like file_inputs.)
    auto hand_counters = [&]() {
      if (block == 0 && tid == 0) {
        if (INC) {
            auto next_t = COLD(int64_t, next_t); auto next_step = COLD(int64_t, next_step);
            if (next_t) *next_t = slot_t + 1;
            if (next_step) *next_step = step64 + 1;
        } else {
            auto t_copy = COLD(int64_t, t_copy); auto step_copy = COLD(int64_t, step_copy);
            if (t_copy) *t_copy = slot_t;
            if (step_copy) *step_copy = step64;
        }
      }
    };
    PSTAMP(12);
    step = (uint32_t)step64;
    file = slot_t < (long)a.slots;                                     // never file past the episode storage
    if (tile < tiles) prepare(tile);                                   // arithmetic only
    PSTAMP(13);
    PSTAMP(1);
    const uint8_t* img = lds_raw;
    const float* tail = reinterpret_cast<const float*>(img + (size_t)tail_piece<PREC>() * 1024);
    float* scratch = reinterpret_cast<float*>(lds_raw + IMAGE_BYTES) + wave * SCRATCH;
    // The wave's 16-row tile through the chain; the K-steps synchronise with the image still streaming in (step_sync).  A wave has
    // AT MOST ONE tile (the host sizes the grid for it: bpa = ceil(tiles per agent / 8)): with a loop over further tiles in the
    // kernel every argument and pointer of the input phase stayed live through the chain -- 118 scalar registers spilled and the
    // vector file full (256, against 150 now).
    auto run_tile = [&]() {
        if (PREC == 2) range_check(amax_tiles<4>(0.f, xk));
        const int b = tile * 16 + m;
        const bool valid = b < N;
        const int bc = valid ? b : N - 1;
        const size_t arow = (size_t)agent * N + bc;                    // agent-major row
        if (first) PSTAMP(2);
        // ---- fc1 + LeakyReLU -----------------------------------------------------------------------------------------------
        f32x4 x1[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) x1[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
        Frag4<PREC> fa, fb;                                            // fragment double buffer of the whole tile chain
#define SSD_LOAD_STEP(c_, f_) do { chunk_ready<PREC, c_>(landed, seen); frag4_load<PREC>(img, c_, lane, f_); } while (0)
        SSD_LOAD_STEP(0, fa);
        SSD_LOAD_STEP(1, fb);
        frag4_mma<PREC>(fa, bh[0], bl[0], x1);
        SSD_LOAD_STEP(2, fa);                                          // the GRU's first step: in flight under fc1's second half
        frag4_mma<PREC>(fb, bh[1], bl[1], x1);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(tail + HT_B1 + 16 * ot + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) x1[ot][r] = leaky(fmaf(x1[ot][r], INV, bias[r]));
        }
        if (first) PSTAMP(3);
        // ---- GRU cell: r, z share one accumulator for the input and the hidden side; n needs both separately -------------------
        float* h_row = a.h + arow * 64;
        f32x4 hp[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) hp[ct] = in.hp[ct];
        f32x4 g[16];                                                   // 0-3 r, 4-7 z, 8-11 i_n, 12-15 h_n
#pragma unroll
        for (int ot = 0; ot < 16; ++ot) g[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
        {   // the six GRU products as 12 K-steps, each step's fragments requested one step ahead
            u32x4 xh[2], xl[2], hh[2], hl[2];
            if (PREC == 2) range_check(amax_tiles<4>(0.f, x1));
            operand<PREC>(x1, XS, xh, xl);
            operand<PREC>(hp, XS, hh, hl);
            SSD_LOAD_STEP(3, fb);   frag4_mma<PREC>(fa, xh[0], xl[0], g);          // K-steps 2 .. 7: input side (r, z, n)
            SSD_LOAD_STEP(4, fa);   frag4_mma<PREC>(fb, xh[1], xl[1], g);
            SSD_LOAD_STEP(5, fb);   frag4_mma<PREC>(fa, xh[0], xl[0], g + 4);
            SSD_LOAD_STEP(6, fa);   frag4_mma<PREC>(fb, xh[1], xl[1], g + 4);
            SSD_LOAD_STEP(7, fb);   frag4_mma<PREC>(fa, xh[0], xl[0], g + 8);
            SSD_LOAD_STEP(8, fa);   frag4_mma<PREC>(fb, xh[1], xl[1], g + 8);
            SSD_LOAD_STEP(9, fb);   frag4_mma<PREC>(fa, hh[0], hl[0], g);          // K-steps 8 .. 13: hidden side
            SSD_LOAD_STEP(10, fa);  frag4_mma<PREC>(fb, hh[1], hl[1], g);
            SSD_LOAD_STEP(11, fb);  frag4_mma<PREC>(fa, hh[0], hl[0], g + 4);
            SSD_LOAD_STEP(12, fa);  frag4_mma<PREC>(fb, hh[1], hl[1], g + 4);
            SSD_LOAD_STEP(13, fb);  frag4_mma<PREC>(fa, hh[0], hl[0], g + 12);
            frag4_mma<PREC>(fb, hh[1], hl[1], g + 12);
            file_inputs(tile);                                         // the whole image has landed: stores may follow
            hand_counters();
        }
        if (first) PSTAMP(4);
        u32x4 f2h[2], f2l[2];                                          // fc2's fragments: in flight under the gate arithmetic
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint8_t* p2 = img + (size_t)fc2_piece<PREC>(0, s2) * 1024 + lane * 16;
            f2h[s2] = *reinterpret_cast<const u32x4*>(p2);
            if (PREC == 2) f2l[s2] = *reinterpret_cast<const u32x4*>(p2 + 2 * 1024);
        }
        f32x4 hn[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const f32x4 bir = *reinterpret_cast<const f32x4*>(tail + HT_BI + 16 * ft + 4 * q), bhr = *reinterpret_cast<const f32x4*>(tail + HT_BH + 16 * ft + 4 * q);
            const f32x4 biz = *reinterpret_cast<const f32x4*>(tail + HT_BI + 64 + 16 * ft + 4 * q), bhz = *reinterpret_cast<const f32x4*>(tail + HT_BH + 64 + 16 * ft + 4 * q);
            const f32x4 bin = *reinterpret_cast<const f32x4*>(tail + HT_BI + 128 + 16 * ft + 4 * q), bhn = *reinterpret_cast<const f32x4*>(tail + HT_BH + 128 + 16 * ft + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rg = sigmoid_fast(fmaf(g[ft][r], INV, bir[r] + bhr[r]));
                const float zg = sigmoid_fast(fmaf(g[4 + ft][r], INV, biz[r] + bhz[r]));
                const float ng = tanh_fast(fmaf(g[8 + ft][r], INV, bin[r]) + rg * fmaf(g[12 + ft][r], INV, bhn[r]));
                hn[ft][r] = (1.f - zg) * ng + zg * hp[ft][r];
            }
            if (valid) stg128(h_row, (size_t)((16 * ft + 4 * q) * 4), hn[ft]);
        }
        if (first) PSTAMP(5);
        // ---- fc2 (advantages + value, padded to 16 outputs) ---------------------------------------------------------------
        f32x4 o2 = f32x4{0.f, 0.f, 0.f, 0.f};
        operand<PREC>(hn, XS, bh, bl);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (PREC == 2) { o2 = mma<PREC>(f2l[s2], bh[s2], o2); o2 = mma<PREC>(f2h[s2], bl[s2], o2); }
            o2 = mma<PREC>(f2h[s2], bh[s2], o2);
        }
        {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(tail + HT_B2 + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) o2[r] = fmaf(o2[r], INV, b2[r]);
        }
        *reinterpret_cast<f32x4*>(scratch + m * 16 + 4 * q) = o2;      // scratch[row][out]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!INC) {
            const int bb = tile * 16 + lane;
            ColdA ca;
            cold_blocks(COLD_BASE, ca, nullptr, nullptr);
            auto out_actions = ca.out_actions; auto p_act = ca.p_act; auto d_actions = ca.d_actions;
            auto out_i32 = ca.out_actions_i32;
            auto q_out = ca.q_out; auto d_onehot = ca.d_onehot;
            if (lane < 16 && bb < N) {
                float av[16];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(scratch + lane * 16 + 4 * k4);
                    av[4 * k4] = t4[0]; av[4 * k4 + 1] = t4[1]; av[4 * k4 + 2] = t4[2]; av[4 * k4 + 3] = t4[3];
                }
                const float val = av[A];
                const uint32_t rq = (uint32_t)(agent * N + bb);                                  // q_out row (agent-major)
                const uint32_t rk = (a.env_id_base + (uint32_t)bb) * (uint32_t)n + (uint32_t)agent;   // exploration key: global env id
                const int act = dueling_pick_bits<AT>(av, val, A, avail_bits, eps, step, a.seed, rk, q_out ? q_out + (size_t)rq * A : (decltype(q_out))nullptr);
                out_actions[(size_t)bb * n + agent] = act;
                if (out_i32) out_i32[(size_t)bb * n + agent] = act;
                if (p_act) p_act[(size_t)bb * n + agent] = act;
                if (d_actions && file) {
                    const size_t sr = ((size_t)bb * a.slots + slot_t) * n + agent;
                    d_actions[sr] = act;
#pragma unroll
                    for (int k = 0; k < A; ++k) d_onehot[sr * A + k] = k == act ? 1.f : 0.f;
                }
            }
        } else {
            if (first) PSTAMP(8);
            const float* w2o = tail + HT_W2O;                          // [E][4]: 3 advantages + value per extra feature
            ColdA ca; ColdB cb;
            cold_blocks(COLD_BASE, ca, &cb, nullptr);
            auto out_actions = ca.out_actions; auto p_inc = cb.p_inc; auto d_actions_inc = cb.d_actions_inc;
            auto q_out = ca.q_out;
            auto recv_out = COLD(uint8_t, recv_out);
            if (first) PSTAMP(9);
            const uint32_t n_magic = (65536u + (uint32_t)n - 1u) / (uint32_t)n;      // it / n for it < 192, n <= 10
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (first && k == 1) PSTAMP(10);
                if (first && k == 2) PSTAMP(11);
                const int it = lane + 64 * k;
                const int row = (int)(((uint32_t)it * n_magic) >> 16), j = it - row * n, bb = tile * 16 + row;
                if (it >= 16 * n || bb >= N) continue;
                const int aj = in.aj[k];
                float f[7];
                f[0] = in.f[k][0] / a.pos_scale; f[1] = in.f[k][1] / a.pos_scale;
#pragma unroll
                for (int e = 2; e < 7; ++e) f[e] = in.f[k][e];
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scratch + row * 16);
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w2o + aj * 4);
                float av[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) av[o] = sc[o] + wa[o];
#pragma unroll
                for (int e = 0; e < 7; ++e) {
                    const f32x4 we = *reinterpret_cast<const f32x4*>(w2o + (A + e) * 4);
#pragma unroll
                    for (int o = 0; o < 4; ++o) av[o] = fmaf(f[e], we[o], av[o]);
                }
                const uint32_t rq = (uint32_t)((agent * N + bb) * n + j);
                const uint32_t rk = ((a.env_id_base + (uint32_t)bb) * (uint32_t)n + (uint32_t)agent) * (uint32_t)n + (uint32_t)j;
                int act = dueling_pick_bits<3>(av, av[3], 3, 0xFFFFFFFFu, eps, step, a.seed, rk, q_out ? q_out + (size_t)rq * 3 : (decltype(q_out))nullptr);
                if (j == agent) act = 0;                               // no self incentive (homophily_controller.py:44-46)
                const size_t pair = ((size_t)bb * n + agent) * n + j;
                out_actions[pair] = act;
                if (p_inc) p_inc[pair] = act;
                if (recv_out) recv_out[((size_t)j * N + bb) * 16 + agent] = (uint8_t)act;      // receiver-major byte (the next env head's input)
                if (d_actions_inc && file) d_actions_inc[(((size_t)bb * a.slots + slot_t) * n + agent) * n + j] = act;
            }
            // once per (env, agent), lanes < 16 (row = lane): this step's outcome of the agent itself, read with the tile's inputs
            const int br = tile * 16 + lane;
            if (lane < 16 && br < N && cb.d_reward) {
                const size_t ea = (size_t)br * n + agent, sr = ((size_t)br * a.slots + slot_t) * n + agent;
                if (file) { cb.d_reward[sr] = in.own[0]; cb.d_clean[sr] = in.own[1]; cb.d_den[sr] = in.own[2]; }
                if (cb.p_rew) cb.p_rew[ea] = in.own[0];
                if (cb.ep_ret) cb.ep_ret[ea] = in.ep + in.own[0];
                if (agent == 0 && cb.d_term && file) cb.d_term[(size_t)br * a.slots + slot_t] = (uint8_t)in.term;
            }
        }
        __builtin_amdgcn_wave_barrier();                               // scratch is reused by the next tile
        if (first) PSTAMP(6);
        first = false;
#undef SSD_LOAD_STEP
    };
    if (tile < tiles) run_tile();
    else hand_counters();
    };   // pass
    const int lane0 = tid & 63;
    {
        const int bia = block - (block / a_entry.bpa) * a_entry.bpa;
        const int tile0 = wave * a_entry.bpa + bia;                    // consecutive tiles go to different CUs
        if constexpr (!LOOP) {
            pass(a_entry, lane0, tile0);
        } else {
            const int stride = a_entry.bpa * WAVES, tiles_all = (a_entry.N + 15) >> 4;
            int tile = tile0;
            do {                                                       // the image is resident from the second pass on (seen == every chunk)
                HeadK al;
                refetch_head_args(al, KOFF);
                int lane = lane0;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(lane));
#endif
                pass(al, lane, tile);
                tile += stride;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+s"(tile));
#endif
            } while (tile < tiles_all);
        }
    }
    {
        const HeadK& a = a_entry;
        const int lane = lane0;
        (void)a; (void)lane;
        PSTAMP(7);
        PSTAMP_REAL(15);
    }
}

// the standalone heads: 8 tiles per workgroup + the loader wave (Cleanup-5 x 4096 envs: 256 tiles per agent -> 32 workgroups per agent, 160 in
// all; measured 4 / 6 / 8 / 11 compute waves: 19.1 / 14.8 / 14.2 / 17.4 us)
#ifndef SSD_HEAD_WAVES
#define SSD_HEAD_WAVES 8
#endif
constexpr int HEAD_WAVES = SSD_HEAD_WAVES;
#ifndef SSD_HEAD_WAVES_LOOP
#define SSD_HEAD_WAVES_LOOP 8
#endif
constexpr int HEAD_WAVES_LOOP = SSD_HEAD_WAVES_LOOP;   // (round 3: 6, when the looped instantiations filled the register file; 160-168 registers now)
constexpr int head_waves(bool loop) { return loop ? HEAD_WAVES_LOOP : HEAD_WAVES; }
// The leading scalar arguments repeat what a compute wave needs for its largest loads (the tile's state and input rows): built
// with -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave (struct arguments are not preloaded).  The looped
// instantiations re-read HeadK from the segment at every pass (refetch_head_args) and ignore them.
constexpr int HEAD_LEAD_BYTES = 2 * 4 + 6 * 8;     // (N | n << 24, bpa, six pointers: 14 dwords, what gfx950 preloads)
template <int INC, int PREC, int AT, int GEN = 0, bool LOOP = false>
__global__ __launch_bounds__((head_waves(LOOP) + 1) * 64) void k_head(uint32_t p_Nn, int p_bpa, float* p_h, float* p_inputs, const int64_t* p_prev_actions,
                                                                      const float* p_prev_reward, const uint8_t* p_recv, const float* p_pos, HeadK a, HeadCold cold_unused) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    if constexpr (LOOP) head_body<INC, PREC, AT, GEN, head_waves(LOOP), LOOP, HEAD_LEAD_BYTES>(a, lds_raw, (int)blockIdx.x);
    else {
        HeadK al = a;
        al.N = (int)(p_Nn & 0xFFFFFFu); al.n = (int)(p_Nn >> 24); al.bpa = p_bpa; al.h = p_h; al.inputs = p_inputs;
        if constexpr (!INC) { al.prev_actions = p_prev_actions; al.prev_reward = p_prev_reward; al.recv = p_recv; al.pos = p_pos; }
        head_body<INC, PREC, AT, GEN, head_waves(LOOP), LOOP, HEAD_LEAD_BYTES, !INC>(al, lds_raw, (int)blockIdx.x);
    }
}

static int chip_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}
static void head_args(const ssd_policy_head* p, HeadK& k, HeadCold& c, int waves = HEAD_WAVES) {
    k.N = p->n_env; k.n = p->n_agents; k.A = p->n_actions; k.inp = p->input_shape;
    k.pos_scale = p->pos_scale; k.seed = p->seed; k.env_id_base = p->env_id_base;
    k.inputs = p->inputs; k.h = p->h; k.weights = static_cast<const uint8_t*>(p->weights); k.avail = p->avail; k.eps = p->epsilon; k.step = p->step;
    k.prev_actions = p->prev_actions; k.prev_reward = p->prev_reward; k.prev_inc = p->prev_actions_inc; k.pos = p->pos; k.orient = p->orient;
    k.actions = p->actions; k.pos_pre = p->pos_pre; k.orient_pre = p->orient_pre; k.reward = p->reward; k.clean = p->clean_num;
    k.den = p->apple_den;
    k.t_index = p->t_index; k.slots = p->t_index ? p->t_slots : 1;
    k.feat_part = p->feat_part; k.feat_bands = p->feat_bands; k.lin_b = p->lin_b;
    k.ep_ret = p->ep_return; k.term = p->terminated;
    k.recv = p->recv_inc; k.avail_bits = p->avail_bits; c.recv_out = p->recv_inc_out;
    c.out_actions = p->out_actions; c.q_out = p->q_out; c.out_actions_i32 = p->out_actions_i32; c.pos_copy = p->pos_copy; c.orient_copy = p->orient_copy;
    c.d_pos = p->dst_pos; c.d_orient = p->dst_orient; c.d_onehot = p->dst_actions_onehot; c.d_reward = p->dst_reward;
    c.d_clean = p->dst_clean_num; c.d_den = p->dst_apple_den; c.d_term = p->dst_terminated;
    c.d_actions = p->dst_actions; c.d_actions_inc = p->dst_actions_inc; c.p_act = p->prev_actions_out; c.p_inc = p->prev_actions_inc_out;
    c.p_rew = p->prev_reward_out; c.ep_ret = p->ep_return; c.next_t = p->next_t_out;
    {   // first tail column of each block, reference order (homophily_controller.py:137-184); input_flags 0 = the shipped set
        const uint32_t fl = p->input_flags ? (p->input_flags & ~SSD_INPUT_EXPLICIT) : (uint32_t)SSD_INPUT_FLAGS_SHIPPED;
        const int width[6] = {k.A, k.n, 1, 1, k.n, 2};
        const uint32_t bit[6] = {SSD_INPUT_LAST_ACTION, SSD_INPUT_AGENT_ID, SSD_INPUT_REWARD, SSD_INPUT_INC_REWARD, SSD_INPUT_DISTANCE,
                                 SSD_INPUT_AGENT_POS};
        uint64_t lay = 0;
        int col = 0;
        for (int f = 0; f < 6; ++f) {
            const int o = (fl & bit[f]) ? col : TAIL_ABSENT;
            if (fl & bit[f]) col += width[f];
            lay |= (uint64_t)(uint8_t)(int8_t)o << (8 * f);
        }
        c.tail_layout = lay;
    }
    c.next_step = p->next_step_out; c.t_copy = p->t_copy_out; c.step_copy = p->step_copy_out;
    c.numeric_err = numeric_err_word();
    PSTAMP_SET(k);
    // workgroups per agent: one 16-row tile per wave while that grid fits the chip (no back edge in the kernel); larger jobs get one
    // workgroup per CU and waves that walk several tiles (the LOOP instantiations)
    const int tiles = (k.N + 15) / 16;
    k.bpa = (tiles + waves - 1) / waves;
    static const bool no_loop = getenv("SSD_HEAD_NO_LOOP") != nullptr;       // experiment: grids larger than the chip instead of looping waves
    if (k.n * k.bpa > chip_cus() && !no_loop) { k.bpa = chip_cus() / k.n; if (k.bpa < 1) k.bpa = 1; }
}
static bool head_loops(const HeadK& k, int waves) { return k.bpa * waves < (k.N + 15) / 16; }
// the standalone heads: one tile per wave while that grid fits the chip (the kernels without a back edge), else the looped instantiation
// (the generic-layout kernels exist as looped instantiations only).  Returns the compute waves per workgroup; `looped` = which kernel.
static int head_plan(const ssd_policy_head* p, HeadK& k, HeadCold& c, bool gen, bool& looped) {
    head_args(p, k, c, gen ? HEAD_WAVES_LOOP : HEAD_WAVES);
    looped = gen || head_loops(k, HEAD_WAVES);
    if (!looped) return HEAD_WAVES;
    head_args(p, k, c, HEAD_WAVES_LOOP);
    return HEAD_WAVES_LOOP;
}

int launch_policy_head(const ssd_policy_head* p, int inc, hipStream_t s) {
    HeadK k;
    HeadCold c;
    const bool gen = !inc && p->input_flags && (p->input_flags & ~SSD_INPUT_EXPLICIT) != (uint32_t)SSD_INPUT_FLAGS_SHIPPED;
    bool looped = false;
    const int waves = head_plan(p, k, c, gen, looped);
    const int prec = p->precision == 1 ? 1 : 2;
    const int bpa = k.bpa;
    const size_t lds = (size_t)head_lds_bytes(waves, prec);
    if (k.A != 9 && k.A != 8) return -3;                               // instantiated for Cleanup (9 actions) and Harvest (8)
    static bool attr_done_dev[64] = {};                               // the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    const void* fns[16] = {reinterpret_cast<const void*>(&k_head<0, 2, 9>), reinterpret_cast<const void*>(&k_head<1, 2, 9>),
                           reinterpret_cast<const void*>(&k_head<0, 1, 9>), reinterpret_cast<const void*>(&k_head<1, 1, 9>),
                           reinterpret_cast<const void*>(&k_head<0, 2, 8>), reinterpret_cast<const void*>(&k_head<1, 2, 8>),
                           reinterpret_cast<const void*>(&k_head<0, 1, 8>), reinterpret_cast<const void*>(&k_head<1, 1, 8>),
                           reinterpret_cast<const void*>(&k_head<0, 2, 9, 0, true>), reinterpret_cast<const void*>(&k_head<1, 2, 9, 0, true>),
                           reinterpret_cast<const void*>(&k_head<0, 1, 9, 0, true>), reinterpret_cast<const void*>(&k_head<1, 1, 9, 0, true>),
                           reinterpret_cast<const void*>(&k_head<0, 2, 8, 0, true>), reinterpret_cast<const void*>(&k_head<1, 2, 8, 0, true>),
                           reinterpret_cast<const void*>(&k_head<0, 1, 8, 0, true>), reinterpret_cast<const void*>(&k_head<1, 1, 8, 0, true>)};
    if (!attr_done_dev[dev]) {
        const size_t l2 = (size_t)head_lds_bytes(HEAD_WAVES, 2);
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess) return -1;
        attr_done_dev[dev] = true;
    }
    const void* gen_fns[4] = {reinterpret_cast<const void*>(&k_head<0, 2, 9, 1, true>), reinterpret_cast<const void*>(&k_head<0, 1, 9, 1, true>),
                              reinterpret_cast<const void*>(&k_head<0, 2, 8, 1, true>), reinterpret_cast<const void*>(&k_head<0, 1, 8, 1, true>)};      // (any grid)
    static bool gen_attr_done_dev[64] = {};
    if (gen && !gen_attr_done_dev[dev]) {
        const size_t l2 = (size_t)head_lds_bytes(HEAD_WAVES, 2);
        for (const void* f : gen_fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess) return -1;
        gen_attr_done_dev[dev] = true;
    }
    if (k.N >= (1 << 24) || k.n >= 256) return -1;
    uint32_t nn = (uint32_t)k.N | ((uint32_t)k.n << 24);
    void* args[10] = {&nn, &k.bpa, &k.h, &k.inputs, &k.prev_actions, &k.prev_reward, &k.recv, &k.pos, &k, &c};
    const void* fn = gen ? gen_fns[(k.A == 8 ? 2 : 0) + (prec == 1 ? 1 : 0)]
                         : fns[(looped ? 8 : 0) + (k.A == 8 ? 4 : 0) + (prec == 1 ? 2 : 0) + (inc ? 1 : 0)];
    if (hipLaunchKernel(fn, dim3(k.n * bpa), dim3((waves + 1) * 64), args, lds, s) != hipSuccess) return -1;
    return 0;
}

// ---- pack: reference-shaped f32 parameters -> the per-agent head image --------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void k_pack_head(ssd_policy_head_params p, uint8_t* image, int32_t* err) {
    constexpr size_t IMAGE_BYTES = SSD_POLICY_IMAGE_BYTES(PREC);
    constexpr float WS = PREC == 2 ? HEAD_WSCALE : 1.f;
    const int agent = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    uint8_t* img = image + (size_t)agent * IMAGE_BYTES;
    if (e < HF_TOT * 512) {
        const int F = e >> 9, lane = (e >> 3) & 63, j = e & 7, q = lane >> 4, m = lane & 15;
        float w = 0.f;
        int piece, term_pieces = 4;                                    // hi term's piece; the lo term sits term_pieces further
        if (F < HF_WI) {
            const int ot = F >> 1, s = F & 1, out = 16 * ot + m, k = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
            if (k < p.fc1_in) w = p.fc1_w[((size_t)agent * p.fc1_in + k) * 64 + out];
            piece = step_piece<PREC>(s, 0, ot);
        } else if (F < HF_FC2) {
            const bool hid = F >= HF_WH;
            const int G = F - (hid ? HF_WH : HF_WI), ot = G >> 1, s = G & 1, out = 16 * ot + m, k = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
            const float* W = (hid ? p.w_h : p.w_i)[out >> 6];
            w = W[((size_t)agent * 64 + k) * 64 + (out & 63)];
            piece = step_piece<PREC>((hid ? 8 : 2) + 2 * (ot >> 2) + s, 0, ot & 3);    // gate g = ot >> 2 (r, z, n), its 4 output tiles
        } else {
            const int s = F - HF_FC2, k = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);   // the h part of fc2: rows 0..63 of [fc2_in, out]
            if (m < p.fc2_out) w = p.fc2_w[((size_t)agent * p.fc2_in + k) * p.fc2_out + m];
            else if (m == p.fc2_out) w = p.fc2_v_w[(size_t)agent * p.fc2_in + k];
            piece = fc2_piece<PREC>(0, s); term_pieces = 2;
        }
        store_term<PREC>(img + ((size_t)piece * 64 + lane) * 16 + 2 * j, w * WS, (size_t)term_pieces * 1024, err);
    }
    if (e < HT_TOT) {
        float v = 0.f;
        if (e < HT_BI) v = p.fc1_b[(size_t)agent * 64 + e];
        else if (e < HT_BH) { const int i = e - HT_BI; v = p.b_i[i >> 6][(size_t)agent * 64 + (i & 63)]; }
        else if (e < HT_B2) { const int i = e - HT_BH; v = p.b_h[i >> 6][(size_t)agent * 64 + (i & 63)]; }
        else if (e < HT_W2O) {
            const int o = e - HT_B2;
            if (o < p.fc2_out) v = p.fc2_b[(size_t)agent * p.fc2_out + o]; else if (o == p.fc2_out) v = p.fc2_v_b[agent];
        } else {
            const int i = e - HT_W2O, ee = i >> 2, o = i & 3;          // pair part of fc2 (inc): rows 64.. of [fc2_in, out]
            if (64 + ee < p.fc2_in) {
                if (o < p.fc2_out) v = p.fc2_w[((size_t)agent * p.fc2_in + 64 + ee) * p.fc2_out + o];
                else if (o == p.fc2_out) v = p.fc2_v_w[(size_t)agent * p.fc2_in + 64 + ee];
            }
        }
        reinterpret_cast<float*>(img + (size_t)tail_piece<PREC>() * 1024)[e] = v;      // the tail follows fc2 in the resident chunk
    }
}

void launch_pack_head(const ssd_policy_head_params* p, int prec, void* image, hipStream_t s) {
    const dim3 grid((HF_TOT * 512 + 255) / 256, p->n_agents);
    int32_t* err = numeric_err_word();
    if (prec == 2) hipLaunchKernelGGL(k_pack_head<2>, grid, dim3(256), 0, s, *p, static_cast<uint8_t*>(image), err);
    else hipLaunchKernelGGL(k_pack_head<1>, grid, dim3(256), 0, s, *p, static_cast<uint8_t*>(image), err);
}

// ===========================================================================================================================
// encoder
// ===========================================================================================================================
// Geometry of one window size.  O output positions per output row, cut into NXT tiles of 8 positions; a conv MFMA tile is
// (2 output channels) x (8 positions) and reads a 10-cell window of the three planes of ONE input row: K = 3 * 8 cells + the 2
// trailing cells of each plane (the "tail" quarter) = 30 of 32.  Two neighbouring position tiles (a pair, XTP pairs per output row)
// are one K-step of the Linear.  The output rows are cut into NB bands of R rows (one workgroup each).
template <int V> struct Geo;
template <> struct Geo<15> { static constexpr int O = 13, CP = 16, NXT = 2, XTP = 1, R = 13, NB = 1; };
template <> struct Geo<31> { static constexpr int O = 29, CP = 32, NXT = 4, XTP = 2, R = 10, NB = 3; };
// (a template parameter BT of the kernels: 4 for 15 x 15 windows at up to 32 768 rows -- Cleanup-5 x 4096: 320 workgroups balance the
// chip better behind the inc heads, k_inc_encode 32.4 -> 30.5 us -- else 5, which moves fewer Linear fragments per row)
constexpr int ENC_BT_MAX = 5;                      // batch tiles (16 rows each) per workgroup: Linear weights are fetched once per 80 rows
#ifndef SSD_ENC_WAVES
#define SSD_ENC_WAVES 8
#endif
constexpr int ENC_WAVES = SSD_ENC_WAVES;                     // 2 per SIMD: one wave's LDS / VALU work overlaps its partner's MFMAs
// LDS record of (batch row, input row): [plane R: CP one-hot bytes][plane G][plane B][tail: NXT x 8 bytes].  tail[k] = cells 8 (k + 1)
// and 8 (k + 1) + 1 of the three planes (+ 2 zero bytes): the 4th K-quarter of position tile k, so that every lane's B operand is
// ONE 8-byte LDS read.  The per-batch-row stride is padded to 8 * odd (mod 256): the 16 rows of a read hit 16 different bank pairs.
template <int V> constexpr int enc_row_bytes() { return 3 * Geo<V>::CP + 8 * Geo<V>::NXT; }
template <int V> constexpr int enc_batch_row_bytes() {
    int b = (Geo<V>::R + 2) * enc_row_bytes<V>();
    while ((b & 255) % 16 != 8) b += 8;
    return b;
}
template <int V, int PREC, int BT = ENC_BT_MAX> constexpr size_t enc_lds_bytes() {
    const size_t work = (size_t)BT * 16 * enc_batch_row_bytes<V>() + (size_t)PREC * 9 * 1024;
    const size_t red = (size_t)ENC_WAVES * BT * 2 * 1024;           // the reduction scratch aliases planes + fragments
    return work > red ? work : red;
}

struct EncK {
    const uint8_t* codes; long code_bytes, env_stride, slot_stride, agent_stride; const int64_t* slot_t;
    int slot_add;                      // the time slot read is *slot_t + slot_add
    int rows, n, agent_major;
    const uint8_t *conv_frags, *lin_frags;
    const float *conv_b, *lin_b;
    float* out; int out_stride;
    float* part;
    float* act;                        // ACT kernels: LeakyReLU(conv) f32 [rows, 6, O, O] (what the learner's backward needs)
    int64_t* slot_t_copy; int64_t* counter_inc;
    int mask_alphabet;                 // bytes are channel masks (1 R, 2 G, 4 B) instead of SSD_OBS_CODE classes
    int layout;                        // SSD_ENCODE_LAYOUT_*: which images conv_frags / lin_frags hold (Toeplitz fragments | class-LUT table + position-major Linear)
    PSTAMP_DECL
};

template <int V, int PREC, bool ACT, int BT>
__device__ __forceinline__ void encode_body(const EncK& a, uint8_t* lds_raw, const int block_x, const int block_y) {
    using G = Geo<V>;
    constexpr int O = G::O, CP = G::CP, XTP = G::XTP, R = G::R;
    constexpr int RB = enc_row_bytes<V>(), PR = enc_batch_row_bytes<V>();   // bytes of an input-row record / of a batch row (this band)
    constexpr int PLANES = BT * 16 * PR, CONV_BYTES = PREC * 9 * 1024;
    constexpr uint32_t ON = PREC == 2 ? 0x3Cu : 0x3Fu;                 // f16 0x3C00 = 1.0;  bf16 0x3F00 = 0.5 (the conv weights carry the 2)
    constexpr float CS = PREC == 2 ? ENC_CSCALE : 1.f, INV = PREC == 2 ? 1.f / (ENC_CSCALE * ENC_LSCALE) : 1.f;
    uint8_t* planes = lds_raw;                                         // [BT * 16 rows][R + 2 input rows][RB]
    uint8_t* cfr = lds_raw + PLANES;                                   // conv fragments [term][s][dy][lane][16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int row0 = block_x * (BT * 16);
    const int band = block_y, y0 = band * R;
    const int Rb = O - y0 < R ? O - y0 : R;                            // output rows of this band
    const long t_off = a.slot_t ? ((long)(*a.slot_t) + a.slot_add) * a.slot_stride : 0;
    PSTAMP(0);
    PSTAMP_REAL(14);
    if (block_x == 0 && block_y == 0 && tid == 0) {
        if (a.slot_t_copy) *a.slot_t_copy = *a.slot_t;
        if (a.counter_inc) *a.counter_inc += 1;
    }
    // ---- stage: conv fragments (global -> LDS), class codes -> one-hot plane bytes ---------------------------------------
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.conv_frags);
        u32x4* dst = reinterpret_cast<u32x4*>(cfr);
        constexpr int NV = CONV_BYTES / 16, PER = (NV + ENC_WAVES * 64 - 1) / (ENC_WAVES * 64);
        u32x4 tmp[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int e = tid + j * ENC_WAVES * 64; if (e < NV) tmp[j] = src[e]; }
        // plane bytes: one item = one window row (V codes at any byte alignment) of a (batch row, band input row): NQ unaligned
        // 16-byte loads (gfx950 global loads take any byte address); an item whose last load would cross the end of the readable
        // bytes is assembled from aligned dwords instead.  Every load of the workgroup is in flight before the first plane byte
        // is written.
        constexpr int NQ = CP / 16, ITEMS = BT * 16 * (R + 2), NT = ENC_WAVES * 64, IPT = (ITEMS + NT - 1) / NT;
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        // (addresses formed as integers: say that they are global, or the loads become FLAT instructions, which count in lgkmcnt too)
        typedef __attribute__((address_space(1))) u32x4_u gl_u32x4_u;
        typedef __attribute__((address_space(1))) uint32_t gl_u32;
        u32x4 cw[IPT][NQ];
        const uintptr_t cbase = reinterpret_cast<uintptr_t>(a.codes), cend = cbase + (uintptr_t)a.code_bytes;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int it = tid + k * NT;
            const int yy = it % (R + 2), r = it / (R + 2);
            const int row = row0 + r, y = y0 + yy;
#pragma unroll
            for (int h = 0; h < NQ; ++h) cw[k][h] = u32x4{0u, 0u, 0u, 0u};
            if (it < ITEMS && row < a.rows && y < V) {
                const int b = row / a.n, i = row - b * a.n;
                const uintptr_t ptr = cbase + (uintptr_t)((long)b * a.env_stride + t_off + (long)i * a.agent_stride + y * V);
                if (ptr + 16 * NQ <= cend) {
#pragma unroll
                    for (int h = 0; h < NQ; ++h) cw[k][h] = *reinterpret_cast<const gl_u32x4_u*>(ptr + 16 * h);
                } else {                                               // the last rows of the buffer: aligned dwords that hold readable bytes
                    const uintptr_t al = ptr & ~(uintptr_t)3, lim = (cend + 3) & ~(uintptr_t)3;
                    uint32_t prev = *reinterpret_cast<const gl_u32*>(al);
#pragma unroll
                    for (int h = 0; h < NQ; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uintptr_t nx = al + 4 * (4 * h + e + 1);
                            const uint32_t next = nx + 4 <= lim ? *reinterpret_cast<const gl_u32*>(nx) : 0u;
                            cw[k][h][e] = __builtin_amdgcn_alignbyte(next, prev, (uint32_t)(ptr & 3));
                            prev = next;
                        }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int it = tid + k * NT;
            if (it >= ITEMS) continue;
            const int yy = it % (R + 2), r = it / (R + 2);
            uint8_t* d = planes + (size_t)r * PR + yy * RB;
            // classes (SSD_OBS_CODE, 0..3): 2 = waste -> R, 1 = apple -> G, 3 = wall / agent -> B (cleanup.py:93-105): with the class
            // bits (b1 b0), B = b1 & b0, R = b1 ^ B, G = b0 ^ B.  Channel masks: plane ch is bit ch.  Cells >= V of the row are cleared.
            u32x4 pl[3][NQ];
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                u32x4 c = cw[k][h];
                if (h == NQ - 1) c[3] &= 0x00FFFFFFu;                  // byte 16 NQ - 1 is cell V (the next row's first cell)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t b0 = c[e] & 0x01010101u, b1 = (c[e] >> 1) & 0x01010101u;
                    if (a.mask_alphabet) {
                        pl[0][h][e] = b0 * ON; pl[1][h][e] = b1 * ON; pl[2][h][e] = ((c[e] >> 2) & 0x01010101u) * ON;
                    } else {
                        const uint32_t bb = b1 & b0;
                        pl[0][h][e] = (b1 ^ bb) * ON; pl[1][h][e] = (b0 ^ bb) * ON; pl[2][h][e] = bb * ON;
                    }
                }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {                        // records are 8-byte aligned (see enc_batch_row_bytes)
                    *reinterpret_cast<u32x2*>(d + ch * CP + 16 * h) = u32x2{pl[ch][h][0], pl[ch][h][1]};
                    *reinterpret_cast<u32x2*>(d + ch * CP + 16 * h + 8) = u32x2{pl[ch][h][2], pl[ch][h][3]};
                }
            }
            // the tail quarters: tail[k] = cells 8 (k + 1), 8 (k + 1) + 1 of the three planes; the last one (cells past the planes) is zero
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                u32x4 tl;
                tl[0] = (pl[0][h][2] & 0xFFFFu) | (pl[1][h][2] << 16); tl[1] = pl[2][h][2] & 0xFFFFu;          // cells 16 h + 8, + 9
                if (h + 1 < NQ) { tl[2] = (pl[0][h + 1][0] & 0xFFFFu) | (pl[1][h + 1][0] << 16); tl[3] = pl[2][h + 1][0] & 0xFFFFu; }   // 16 h + 16, + 17
                else { tl[2] = 0u; tl[3] = 0u; }
                *reinterpret_cast<u32x2*>(d + 3 * CP + 16 * h) = u32x2{tl[0], tl[1]};
                *reinterpret_cast<u32x2*>(d + 3 * CP + 16 * h + 8) = u32x2{tl[2], tl[3]};
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int e = tid + j * ENC_WAVES * 64; if (e < NV) dst[e] = tmp[j]; }
    }
    __syncthreads();
    PSTAMP(1);
    // ---- units (output row y, position-tile pair xtp, output-channel pair s) of this band, a contiguous range per wave ---------
    f32x4 accl[BT][2];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) { accl[bt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[bt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int U = Rb * XTP * 3;
    const int u_begin = (wave * U) / ENC_WAVES, u_end = ((wave + 1) * U) / ENC_WAVES;
    const uint8_t* my_b = planes + (size_t)m * PR + q * CP;            // quarter q < 3: plane q, 8 cells; quarter 3: the tail
    // Linear A fragments of a unit (global, L2-resident) are requested one unit ahead of their use
    auto load_la = [&](int u, u32x4 (&la)[2][PREC]) {
        const int s = u % 3, xtp = (u / 3) % XTP, yl = u / (3 * XTP);
        const size_t gu = (size_t)((y0 + yl) * XTP + xtp) * 3 + s;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < PREC; ++t)
                la[mt][t] = *reinterpret_cast<const u32x4*>(a.lin_frags + (((gu * 2 + mt) * PREC + t) * 64 + lane) * 16);
    };
    u32x4 la[2][PREC], la_next[2][PREC];
    if (u_begin < u_end) load_la(u_begin, la_next);
    // this lane's conv bias for each channel pair, fetched once: a load per unit at the top of the loop was waited for with vmcnt(0) --
    // a full L2 latency per unit, and the Linear fragments requested a unit ahead were waited for with it
    float cbq[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) cbq[s3] = a.conv_b[2 * s3 + (q >> 1)] * CS;
#pragma unroll 1
    for (int u = u_begin; u < u_end; ++u) {
        const int s = u % 3, xtp = (u / 3) % XTP, yl = u / (3 * XTP);  // yl: output row inside the band
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < PREC; ++t) la[mt][t] = la_next[mt][t];
        if (u + 1 < u_end) load_la(u + 1, la_next);
        f32x4 accc[2][BT];                                             // [position tile of the pair][batch tile]: rows (o2 = q >> 1, 8 positions)
        {
            const float bq = s == 0 ? cbq[0] : (s == 1 ? cbq[1] : cbq[2]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) { accc[0][bt] = f32x4{bq, bq, bq, bq}; accc[1][bt] = f32x4{bq, bq, bq, bq}; }
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            u32x4 af[PREC];
#pragma unroll
            for (int t = 0; t < PREC; ++t) af[t] = *reinterpret_cast<const u32x4*>(cfr + ((size_t)((t * 3 + s) * 3 + dy) * 64 + lane) * 16);
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const uint8_t* bp = my_b + (yl + dy) * RB + 8 * (2 * xtp + tx);
                u32x4 bfr[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    const u32x2 w = *reinterpret_cast<const u32x2*>(bp + bt * 16 * PR);
                    bfr[bt][0] = __builtin_amdgcn_perm(0u, w[0], 0x010C000Cu); bfr[bt][1] = __builtin_amdgcn_perm(0u, w[0], 0x030C020Cu);
                    bfr[bt][2] = __builtin_amdgcn_perm(0u, w[1], 0x010C000Cu); bfr[bt][3] = __builtin_amdgcn_perm(0u, w[1], 0x030C020Cu);
                }
#pragma unroll
                for (int t = PREC - 1; t >= 0; --t)
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) accc[tx][bt] = mma<PREC>(af[t], bfr[bt], accc[tx][bt]);
            }
        }
        // LeakyReLU (positively homogeneous: the scale CS rides through), split, Linear K-step of this unit (the two output tiles'
        // accumulation chains are interleaved)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            float v[8];
            if constexpr (ACT || PREC != 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = leaky(accc[j >> 2][bt][j & 3]);
            }
            if constexpr (ACT) {                                       // the training forward keeps the conv activations
                const int row = row0 + bt * 16 + m;
                if (row < a.rows) {
                    float* ar = a.act + ((size_t)row * 6 + 2 * s + (q >> 1)) * (O * O) + (y0 + yl) * O;
                    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte store at any dword address
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx) {                   // this lane's 4 consecutive positions of each tile
                        const int x0 = 8 * (2 * xtp + tx) + 4 * (q & 1);
                        if (x0 + 3 < O) {
                            *reinterpret_cast<f32x4_u*>(ar + x0) = f32x4_u{v[4 * tx] * (1.f / CS), v[4 * tx + 1] * (1.f / CS), v[4 * tx + 2] * (1.f / CS),
                                                                           v[4 * tx + 3] * (1.f / CS)};
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (x0 + r < O) ar[x0 + r] = v[4 * tx + r] * (1.f / CS);
                        }
                    }
                }
            }
            u32x4 xh, xl;
            if constexpr (ACT || PREC != 2) split8<PREC>(v, xh, xl);
            else leaky_split8(accc[0][bt], accc[1][bt], xh, xl);
            if (PREC == 2) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accl[bt][mt] = mma<PREC>(la[mt][PREC - 1], xh, accl[bt][mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accl[bt][mt] = mma<PREC>(la[mt][0], xl, accl[bt][mt]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) accl[bt][mt] = mma<PREC>(la[mt][0], xh, accl[bt][mt]);
        }
    }
    PSTAMP(2);
    // ---- add the waves' partial sums in a fixed order (deterministic), finish ------------------------------------------------------
    __syncthreads();                                                   // every wave is done with the planes: reuse them
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw);                    // [wave][bt][mt][lane]
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) red[((wave * BT + bt) * 2 + mt) * 64 + lane] = accl[bt][mt];
    __syncthreads();
    for (int it = tid; it < BT * 2 * 64; it += ENC_WAVES * 64) {
        const int l = it & 63, mt = (it >> 6) & 1, bt = it >> 7;
        f32x4 sum = red[((0 * BT + bt) * 2 + mt) * 64 + l];
#pragma unroll
        for (int w = 1; w < ENC_WAVES; ++w) sum += red[((w * BT + bt) * 2 + mt) * 64 + l];
        const int row = row0 + bt * 16 + (l & 15), f0 = 16 * mt + 4 * (l >> 4);
        if (row < a.rows) {
            const int b = row / a.n, i = row - b * a.n;
            const size_t orow = a.agent_major ? (size_t)i * (a.rows / a.n) + b : (size_t)row;
            if (a.part) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = sum[r] * INV;
                *reinterpret_cast<f32x4*>(a.part + ((size_t)band * a.rows + orow) * 32 + f0) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a.out[orow * a.out_stride + f0 + r] = leaky(fmaf(sum[r], INV, a.lin_b[f0 + r]));
            }
        }
    }
    PSTAMP(3);
    PSTAMP_REAL(15);
}


// ---------------------------------------------------------------------------------------------------------------------------
// encode_body_lut: the same encoder with the CONVOLUTION AS A TABLE SUM (SSD_ENCODE_LAYOUT_LUT; rollout / no-gradient launches).
// A window cell is one of four classes and lights at most one colour plane at 255/256 (cleanup.py:93-105), so for an output position
// the three taps of input row dy contribute, for all six channels at once, a value that depends only on the three classes under them:
//   conv[c](y, x) = sum_dy T[dy][class(y + dy, x) + 4 class(y + dy, x + 1) + 16 class(y + dy, x + 2)][c],   T[0] carries the bias
// -- 3 x 64 entries of 6 f32 built by ssd_policy_pack_encoder_lut (exact f32 arithmetic: no split of the conv weights, no Toeplitz
// zeros, no matrix-core work for the conv at all; round 3 executed 468 conv MFMAs per 16-row tile, 9 useful of 30 K entries each).
// The window rows are kept in LDS as packed 2-bit classes (one dword per 15-cell row, two per 31-cell row), the table as 4.5 KiB.
// Lane (q, m) evaluates position p = 4 s + q of batch row m for K-step s: its 6 channel values (+ 2 zeros) after LeakyReLU and the
// split ARE its B operand of the Linear's K-step s (k = 8 q + j <-> position 4 s + q, channel j; lin image in that order), so the
// Linear keeps its three products per term pair: 43 K-steps x 2 output tiles x 3 = 258 MFMAs per 16-row tile (15 x 15 windows)
// instead of 702.  K-steps are dealt to the 8 waves as contiguous ranges and the partial sums added in wave order as before.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int LUT_ENTRY_F = 6, LUT_DY_F = 64 * LUT_ENTRY_F, LUT_TABLE_BYTES = SSD_ENCODE_LUT_TABLE_BYTES;      // [3][64][6] f32
static_assert(LUT_TABLE_BYTES == 3 * LUT_DY_F * 4, "table size");
template <int V> constexpr int lut_band_rows(int band) { return Geo<V>::O - band * Geo<V>::R < Geo<V>::R ? Geo<V>::O - band * Geo<V>::R : Geo<V>::R; }
template <int V> constexpr int lut_band_ksteps(int band) { return (lut_band_rows<V>(band) * Geo<V>::O + 3) / 4; }
template <int V> constexpr int lut_band_base(int band) { int b = 0; for (int k = 0; k < band; ++k) b += lut_band_ksteps<V>(k); return b; }
static_assert(lut_band_base<15>(1) == SSD_ENCODE_LUT_KSTEPS(15) && lut_band_base<31>(3) == SSD_ENCODE_LUT_KSTEPS(31), "K-steps of the Linear image");
template <int V> constexpr int lut_row_dwords() { return V == 15 ? 1 : 2; }
template <int V> constexpr int lut_batch_row_dwords() { int d = (Geo<V>::R + 3) * lut_row_dwords<V>(); return d | 1; }      // odd: the 16 rows of a read hit 16 banks
template <int V, int PREC, int BT> constexpr size_t enc_lut_lds_bytes() {
    const size_t work = (size_t)BT * 16 * lut_batch_row_dwords<V>() * 4 + 16 + LUT_TABLE_BYTES;
    const size_t red = (size_t)ENC_WAVES * BT * 2 * 1024;
    return work > red ? work : red;
}
// 4 cells as bytes (classes 0..3) -> 8 bits
__device__ __forceinline__ uint32_t pack4x2(uint32_t c) {
    uint32_t t = (c | (c >> 6)) & 0x000F000Fu;
    return (t | (t >> 12)) & 0xFFu;
}

template <int V, int PREC, int BT>
__device__ __forceinline__ void encode_body_lut(const EncK& a, uint8_t* lds_raw, const int block_x, const int block_y) {
    using G = Geo<V>;
    constexpr int O = G::O, CP = G::CP, R = G::R;
    constexpr int WPR = lut_row_dwords<V>(), PRW = lut_batch_row_dwords<V>();
    constexpr int PACKED = (BT * 16 * PRW * 4 + 15) & ~15;
    constexpr float INV = PREC == 2 ? 1.f / (ENC_CSCALE * ENC_LSCALE) : 1.f;
    uint32_t* packed = reinterpret_cast<uint32_t*>(lds_raw);          // [BT * 16 rows][PRW dwords]: input rows of the band, 2 bits per cell
    float* table = reinterpret_cast<float*>(lds_raw + PACKED);         // [3][64][6]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int row0 = block_x * (BT * 16);
    const int band = block_y, y0 = band * R;
    const int Rb = O - y0 < R ? O - y0 : R;                            // output rows of this band
    // Every argument the staging reads, requested in ONE batch (by-value arguments are fetched where they are first used: six
    // dependent scalar-memory trips before the first code byte is requested otherwise), then the time slot -- read ONCE.
    asm volatile("" :: "s"(a.codes), "s"(a.code_bytes), "s"(a.env_stride), "s"(a.slot_stride), "s"(a.agent_stride), "s"(a.slot_t), "s"(a.slot_add),
                       "s"(a.rows), "s"(a.n), "s"(a.conv_frags), "s"(a.lin_frags), "s"(a.mask_alphabet), "s"(a.slot_t_copy), "s"(a.counter_inc));
    int64_t slot_now = 0;                                              // (a scalar load: the pointer is a kernel argument)
    if (a.slot_t) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(slot_now) : "s"(a.slot_t) : "memory");
#endif
    }
    const long t_off = a.slot_t ? ((long)slot_now + a.slot_add) * a.slot_stride : 0;
    PSTAMP(0);
    PSTAMP_REAL(14);
    if (block_x == 0 && block_y == 0 && tid == 0) {
        if (a.slot_t_copy) *a.slot_t_copy = slot_now;
        if (a.counter_inc) *a.counter_inc += 1;
    }
    // ---- stage: the table (global -> LDS), class codes -> packed rows ------------------------------------------------------------
    {
        constexpr int NV = LUT_TABLE_BYTES / 16;
        const u32x4* src = reinterpret_cast<const u32x4*>(a.conv_frags);
        u32x4 tv = {0u, 0u, 0u, 0u};
        if (tid < NV) tv = src[tid];
        constexpr int NQ = CP / 16, ITEMS = BT * 16 * (R + 2), NT = ENC_WAVES * 64, IPT = (ITEMS + NT - 1) / NT;
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        typedef __attribute__((address_space(1))) u32x4_u gl_u32x4_u;
        typedef __attribute__((address_space(1))) uint32_t gl_u32;
        u32x4 cw[IPT][NQ];
        const uintptr_t cbase = reinterpret_cast<uintptr_t>(a.codes), cend = cbase + (uintptr_t)a.code_bytes;
        // every item's 16-byte reads first (predicated, no use): written as `if (inside) load; else tail path` per item, each item's
        // loads are waited for at the join before the next item's are requested.  The tail path (the buffer's last rows, where a
        // 16-byte read would cross the end) runs afterwards, for the workgroup that holds them.
        uintptr_t ptrs[IPT];
        bool tail[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int it = tid + k * NT;
            const int yy = it % (R + 2), r = it / (R + 2);
            const int row = row0 + r, y = y0 + yy;
#pragma unroll
            for (int h = 0; h < NQ; ++h) cw[k][h] = u32x4{0u, 0u, 0u, 0u};
            ptrs[k] = 0; tail[k] = false;
            if (it < ITEMS && row < a.rows && y < V) {
                const int b = row / a.n, i = row - b * a.n;
                const uintptr_t ptr = cbase + (uintptr_t)((long)b * a.env_stride + t_off + (long)i * a.agent_stride + y * V);
                ptrs[k] = ptr;
                if (ptr + 16 * NQ <= cend) {
#pragma unroll
                    for (int h = 0; h < NQ; ++h) cw[k][h] = *reinterpret_cast<const gl_u32x4_u*>(ptr + 16 * h);
                } else tail[k] = true;
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            if (tail[k]) {                                             // the last rows of the buffer: aligned dwords that hold readable bytes
                const uintptr_t ptr = ptrs[k];
                const uintptr_t al = ptr & ~(uintptr_t)3, lim = (cend + 3) & ~(uintptr_t)3;
                uint32_t prev = *reinterpret_cast<const gl_u32*>(al);
#pragma unroll
                for (int h = 0; h < NQ; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uintptr_t nx = al + 4 * (4 * h + e + 1);
                        const uint32_t next = nx + 4 <= lim ? *reinterpret_cast<const gl_u32*>(nx) : 0u;
                        cw[k][h][e] = __builtin_amdgcn_alignbyte(next, prev, (uint32_t)(ptr & 3));
                        prev = next;
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int it = tid + k * NT;
            if (it >= ITEMS) continue;
            const int yy = it % (R + 2), r = it / (R + 2);
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                u32x4 c = cw[k][h];
                if (h == NQ - 1) c[3] &= 0x00FFFFFFu;                  // byte 16 NQ - 1 is cell V (the next row's first cell)
                uint32_t word = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t cls = c[e] & 0x03030303u;                 // SSD_OBS_CODE classes 0..3
                    if (a.mask_alphabet) {                             // channel masks 1 R / 2 G / 4 B -> classes 2 / 1 / 3
                        const uint32_t b0 = c[e] & 0x01010101u, b1 = (c[e] >> 1) & 0x01010101u, b2 = (c[e] >> 2) & 0x01010101u;
                        cls = (b1 | b2) | ((b0 | b2) << 1);
                    }
                    word |= pack4x2(cls) << (8 * e);
                }
                packed[r * PRW + yy * WPR + h] = word;
            }
        }
        if (tid < NV) reinterpret_cast<u32x4*>(table)[tid] = tv;
    }
    __syncthreads();
    PSTAMP(1);
    // ---- K-steps of this band, a contiguous range per wave -------------------------------------------------------------------------
    f32x4 accl[BT][2];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) { accl[bt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[bt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int KS = (Rb * O + 3) >> 2;
    int ks_base = 0;                                                   // first K-step of this band in the Linear image
#pragma unroll
    for (int k = 0; k < G::NB; ++k) ks_base += k < band ? lut_band_ksteps<V>(k) : 0;
    const int s_begin = (wave * KS) / ENC_WAVES, s_end = ((wave + 1) * KS) / ENC_WAVES;
    auto load_la = [&](int sl, u32x4 (&la)[2][PREC]) {
        const size_t gs = (size_t)(ks_base + sl);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < PREC; ++t)
                la[mt][t] = *reinterpret_cast<const u32x4*>(a.lin_frags + (((gs * 2 + mt) * PREC + t) * 64 + lane) * 16);
    };
    u32x4 la[2][PREC], la_next[2][PREC];
    if (s_begin < s_end) load_la(s_begin, la_next);
    const uint32_t* my_rows = packed + m * PRW;
#ifndef SSD_LUT_UNROLL
#define SSD_LUT_UNROLL 1
#endif
#pragma unroll SSD_LUT_UNROLL
    for (int sl = s_begin; sl < s_end; ++sl) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < PREC; ++t) la[mt][t] = la_next[mt][t];
        if (sl + 1 < s_end) load_la(sl + 1, la_next);
        // this lane's position of the K-step: p = 4 sl + q -> (output row yl of the band, column x); positions past the band read row 0
        // (their Linear weights are zero)
        const int p0 = 4 * sl, yl0 = p0 / O, x0 = p0 - yl0 * O;        // (scalar)
        int x = x0 + q, yl = yl0;
        if (x >= O) { x -= O; yl += 1; }
        if (yl >= Rb) { yl = 0; x = 0; }
        const uint32_t* rp = my_rows + yl * WPR;
        const int sh = 2 * x;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            const uint32_t* rb = rp + bt * 16 * PRW;
            int idx[3];
            if constexpr (WPR == 1) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) idx[dy] = (int)((rb[dy] >> sh) & 63u);
            } else {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const uint64_t w = (uint64_t)rb[2 * dy] | ((uint64_t)rb[2 * dy + 1] << 32);
                    idx[dy] = (int)((w >> sh) & 63u);
                }
            }
            f32x2 sums[3];
            {
                const f32x2* t0 = reinterpret_cast<const f32x2*>(table + idx[0] * LUT_ENTRY_F);
                const f32x2* t1 = reinterpret_cast<const f32x2*>(table + LUT_DY_F + idx[1] * LUT_ENTRY_F);
                const f32x2* t2 = reinterpret_cast<const f32x2*>(table + 2 * LUT_DY_F + idx[2] * LUT_ENTRY_F);
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) sums[c2] = (t0[c2] + t1[c2]) + t2[c2];
            }
            u32x4 xh, xl;
            if constexpr (PREC == 2) {
                leaky_split6(sums, xh, xl);
            } else {
                float v[8];
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) { v[2 * c2] = leaky(sums[c2].x); v[2 * c2 + 1] = leaky(sums[c2].y); }
                v[6] = 0.f; v[7] = 0.f;
                split8<PREC>(v, xh, xl);
            }
            if (PREC == 2) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accl[bt][mt] = mma<PREC>(la[mt][PREC - 1], xh, accl[bt][mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accl[bt][mt] = mma<PREC>(la[mt][0], xl, accl[bt][mt]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) accl[bt][mt] = mma<PREC>(la[mt][0], xh, accl[bt][mt]);
        }
    }
    PSTAMP(2);
    // ---- add the waves' partial sums in a fixed order (deterministic), finish ------------------------------------------------------
    __syncthreads();                                                   // every wave is done with the packed rows and the table: reuse them
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw);                    // [wave][bt][mt][lane]
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) red[((wave * BT + bt) * 2 + mt) * 64 + lane] = accl[bt][mt];
    __syncthreads();
    for (int it = tid; it < BT * 2 * 64; it += ENC_WAVES * 64) {
        const int l = it & 63, mt = (it >> 6) & 1, bt = it >> 7;
        f32x4 sum = red[((0 * BT + bt) * 2 + mt) * 64 + l];
#pragma unroll
        for (int w = 1; w < ENC_WAVES; ++w) sum += red[((w * BT + bt) * 2 + mt) * 64 + l];
        const int row = row0 + bt * 16 + (l & 15), f0 = 16 * mt + 4 * (l >> 4);
        if (row < a.rows) {
            const int b = row / a.n, i = row - b * a.n;
            const size_t orow = a.agent_major ? (size_t)i * (a.rows / a.n) + b : (size_t)row;
            if (a.part) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = sum[r] * INV;
                *reinterpret_cast<f32x4*>(a.part + ((size_t)band * a.rows + orow) * 32 + f0) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a.out[orow * a.out_stride + f0 + r] = leaky(fmaf(sum[r], INV, a.lin_b[f0 + r]));
            }
        }
    }
    PSTAMP(3);
    PSTAMP_REAL(15);
}

template <int V, int PREC, bool ACT, int BT>
__global__ __launch_bounds__(ENC_WAVES * 64) void k_encode(EncK a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    encode_body<V, PREC, ACT, BT>(a, lds_raw, (int)blockIdx.x, (int)blockIdx.y);
}
template <int V, int PREC, int BT>
__global__ __launch_bounds__(ENC_WAVES * 64) void k_encode_lut(EncK a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    encode_body_lut<V, PREC, BT>(a, lds_raw, (int)blockIdx.x, (int)blockIdx.y);
}
static int enc_bt(int V, int rows) {
    static const char* force = getenv("SSD_ENC_BT");                  // diagnostics: force the batch tiles per workgroup (4 | 5) for 15 x 15 windows
    if (force && V == 15) return force[0] == '4' ? 4 : 5;
    return (V == 15 && rows <= 32768) ? 4 : 5;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_inc_encode: the inc head of timestep t and the encoder of timestep t + 1 as ONE launch (pipelined rollout).  Both follow the env
// step of t and share no data: the inc head reads the input rows of t (one buffer of a pair), the encoder reads the observation the
// env step just stored in slot t + 1 and writes the other buffer.  Either kernel fills the register file of a CU with one workgroup,
// so nothing co-resides -- what is saved is one launch-to-launch floor per timestep (3.1 us of 64), and the second half of the grid
// starts on each CU the moment its head workgroup retires.  Workgroups [0, heads) run head_body, the rest encode_body with the
// (x, band) index unfolded; EncK sits behind the two head arguments (the heads' cold-argument offsets are unchanged).
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef SSD_FUSED_WAVES
#define SSD_FUSED_WAVES SSD_ENC_WAVES
#endif
constexpr int FUSED_WAVES = SSD_FUSED_WAVES;    // k_inc_encode: one block size for both bodies (waves past ENC_WAVES leave an encoder workgroup at once)
static_assert(FUSED_WAVES >= ENC_WAVES, "the encoder body needs its waves");
template <int PREC, int AT, int V, bool LOOP, int BT, bool LUT = false>
__global__ __launch_bounds__(FUSED_WAVES * 64) void k_inc_encode(int heads, int enc_groups, int p_N, int p_n, int p_bpa, int p_pad, float* p_h, float* p_inputs,
                                                                 const uint8_t* p_codes, const int64_t* p_slot_t, HeadK a, HeadCold cold_unused, EncK e) {
    // (leading scalars: which body a workgroup runs, then what each body's first requests need -- see k_head)
    constexpr int LEAD = 6 * 4 + 4 * 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
#ifndef SSD_ENC_FIRST
#define SSD_ENC_FIRST 0
#endif
    const int total = (int)gridDim.x;
    const int b = SSD_ENC_FIRST ? ((int)blockIdx.x + heads) % total : (int)blockIdx.x;      // (which body the first workgroups run)
    if (b < heads) {
        if constexpr (LOOP) head_body<1, PREC, AT, 0, FUSED_WAVES - 1, LOOP, LEAD>(a, lds_raw, b);
        else {
            HeadK al = a;
            al.N = p_N; al.n = p_n; al.bpa = p_bpa; al.h = p_h; al.inputs = p_inputs;
            head_body<1, PREC, AT, 0, FUSED_WAVES - 1, LOOP, LEAD>(al, lds_raw, b);    // 7 compute waves + the loader
        }
    } else {
        const int i = b - heads, by = i / enc_groups;
        if (FUSED_WAVES > ENC_WAVES && (int)threadIdx.x >= ENC_WAVES * 64) return;
        EncK el = e;
        el.codes = p_codes; el.slot_t = p_slot_t;
        if constexpr (LUT) encode_body_lut<V, PREC, BT>(el, lds_raw, i - by * enc_groups, by);
        else encode_body<V, PREC, false, BT>(el, lds_raw, i - by * enc_groups, by);
    }
}

template <int V, int PREC, bool ACT, int BT>
static int launch_encode_bt(const EncK& k, hipStream_t s) {
    using G = Geo<V>;
    constexpr size_t lds = enc_lds_bytes<V, PREC, BT>();
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode<V, PREC, ACT, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        done[dev] = true;
    }
    const int groups = (k.rows + BT * 16 - 1) / (BT * 16);
    hipLaunchKernelGGL((k_encode<V, PREC, ACT, BT>), dim3(groups, G::NB), dim3(ENC_WAVES * 64), lds, s, k);
    return 0;
}
template <int V, int PREC, bool ACT>
static int launch_encode_t(const EncK& k, hipStream_t s) {
    if constexpr (V == 15) { if (enc_bt(V, k.rows) == 4) return launch_encode_bt<V, PREC, ACT, 4>(k, s); }
    return launch_encode_bt<V, PREC, ACT, 5>(k, s);
}

template <int V, int PREC, int BT>
static int launch_encode_lut_bt(const EncK& k, hipStream_t s) {
    using G = Geo<V>;
    constexpr size_t lds = enc_lut_lds_bytes<V, PREC, BT>();
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_lut<V, PREC, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        done[dev] = true;
    }
    const int groups = (k.rows + BT * 16 - 1) / (BT * 16);
    hipLaunchKernelGGL((k_encode_lut<V, PREC, BT>), dim3(groups, G::NB), dim3(ENC_WAVES * 64), lds, s, k);
    return 0;
}
template <int V, int PREC>
static int launch_encode_lut_t(const EncK& k, hipStream_t s) {
    if constexpr (V == 15) { if (enc_bt(V, k.rows) == 4) return launch_encode_lut_bt<V, PREC, 4>(k, s); }
    return launch_encode_lut_bt<V, PREC, 5>(k, s);
}

static void encode_args(const ssd_policy_encode_args* p, EncK& k) {
    k.codes = p->codes; k.code_bytes = (long)p->code_bytes; k.env_stride = (long)p->env_stride; k.slot_stride = (long)p->slot_stride;
    k.agent_stride = (long)p->agent_stride; k.slot_t = p->slot_t; k.rows = p->rows; k.n = p->n_agents; k.agent_major = p->agent_major;
    k.conv_frags = static_cast<const uint8_t*>(p->conv_frags); k.lin_frags = static_cast<const uint8_t*>(p->lin_frags);
    k.conv_b = p->conv_b; k.lin_b = p->lin_b; k.out = p->out; k.out_stride = p->out_stride; k.part = p->part; k.act = p->act;
    k.slot_t_copy = p->slot_t_copy; k.counter_inc = p->counter_inc; k.mask_alphabet = p->alphabet == SSD_CODE_CHANNEL_MASK;
    k.slot_add = p->slot_add; k.layout = p->layout;
    PSTAMP_SET(k);
}

template <int PREC, int AT, int V, bool LOOP, int BT, bool LUT>
static int launch_inc_encode_bt(HeadK& k, HeadCold& c, EncK& e, hipStream_t s) {
    const size_t lh = (size_t)head_lds_bytes(FUSED_WAVES - 1, PREC), le = LUT ? enc_lut_lds_bytes<V, PREC, BT>() : enc_lds_bytes<V, PREC, BT>();
    const size_t lds = lh > le ? lh : le;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    const void* fn = reinterpret_cast<const void*>(&k_inc_encode<PREC, AT, V, LOOP, BT, LUT>);
    if (!done[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        done[dev] = true;
    }
    int heads = k.n * k.bpa, groups = (e.rows + BT * 16 - 1) / (BT * 16);
    int pad = 0;
    void* args[13] = {&heads, &groups, &k.N, &k.n, &k.bpa, &pad, &k.h, &k.inputs, &e.codes, &e.slot_t, &k, &c, &e};
    if (hipLaunchKernel(fn, dim3(heads + groups * Geo<V>::NB), dim3(FUSED_WAVES * 64), args, lds, s) != hipSuccess) return -1;
    return 0;
}
template <int PREC, int AT, int V, bool LOOP>
static int launch_inc_encode_t(HeadK& k, HeadCold& c, EncK& e, hipStream_t s) {
    if (e.layout == SSD_ENCODE_LAYOUT_LUT) {
        if constexpr (V == 15) { if (enc_bt(V, e.rows) == 4) return launch_inc_encode_bt<PREC, AT, V, LOOP, 4, true>(k, c, e, s); }
        return launch_inc_encode_bt<PREC, AT, V, LOOP, 5, true>(k, c, e, s);
    }
    if constexpr (V == 15) { if (enc_bt(V, e.rows) == 4) return launch_inc_encode_bt<PREC, AT, V, LOOP, 4, false>(k, c, e, s); }
    return launch_inc_encode_bt<PREC, AT, V, LOOP, 5, false>(k, c, e, s);
}

// How a head launch of (n_env, n_agents) is cut: workgroups per agent, compute waves per workgroup, and the number of 16-row tiles the
// busiest wave walks (1 = the kernel without a back edge; > 1 = the LOOP instantiation).  fused: the inc head inside k_inc_encode.
void policy_head_plan(int n_env, int n_agents, int fused, int* wg_per_agent, int* waves_out, int* tiles_per_wave) {
    HeadK k = {};
    k.N = n_env; k.n = n_agents;
    const int tiles = (n_env + 15) / 16;
    int waves = fused ? FUSED_WAVES - 1 : HEAD_WAVES;
    auto size = [&](int w) {
        k.bpa = (tiles + w - 1) / w;
        static const bool no_loop = getenv("SSD_HEAD_NO_LOOP") != nullptr;
        if (k.n * k.bpa > chip_cus() && !no_loop) { k.bpa = chip_cus() / k.n; if (k.bpa < 1) k.bpa = 1; }
    };
    size(waves);
    if (!fused && head_loops(k, waves)) { waves = HEAD_WAVES_LOOP; size(waves); }
    *wg_per_agent = k.bpa; *waves_out = waves;
    *tiles_per_wave = (tiles + k.bpa * waves - 1) / (k.bpa * waves);
}

// inc head (timestep t) + encoder (timestep t + 1) as one launch; -2: no instance for this window size, -3: for this action count
int launch_policy_inc_encode(const ssd_policy_head* ph, const ssd_policy_encode_args* pe, hipStream_t s) {
    HeadK k;
    HeadCold c;
    EncK e;
    head_args(ph, k, c, FUSED_WAVES - 1);
    encode_args(pe, e);
    const int prec = ph->precision == 1 ? 1 : 2, V = pe->view_edge;
    if (k.A != 9 && k.A != 8) return -3;
    if (V != 15 && V != 31) return -2;
    const bool loops = head_loops(k, FUSED_WAVES - 1);
#define SSD_IE(P_, A_, V_) if (prec == P_ && k.A == A_ && V == V_) return loops ? launch_inc_encode_t<P_, A_, V_, true>(k, c, e, s) : launch_inc_encode_t<P_, A_, V_, false>(k, c, e, s)
    SSD_IE(2, 9, 15); SSD_IE(2, 9, 31); SSD_IE(2, 8, 15); SSD_IE(2, 8, 31);
    SSD_IE(1, 9, 15); SSD_IE(1, 9, 31); SSD_IE(1, 8, 15); SSD_IE(1, 8, 31);
#undef SSD_IE
    return -2;
}

int launch_policy_encode(const ssd_policy_encode_args* p, hipStream_t s) {
    EncK k;
    encode_args(p, k);
    const int prec = p->precision == 1 ? 1 : 2;
    if (p->act) {       // the learner's forward (also emits LeakyReLU(conv)): f32-equivalent, or the labelled bf16 variant
        if (p->view_edge == 15) return prec == 2 ? launch_encode_t<15, 2, true>(k, s) : launch_encode_t<15, 1, true>(k, s);
        if (p->view_edge == 31) return prec == 2 ? launch_encode_t<31, 2, true>(k, s) : launch_encode_t<31, 1, true>(k, s);
        return -2;
    }
    if (p->layout == SSD_ENCODE_LAYOUT_LUT) {
        if (p->view_edge == 15) return prec == 2 ? launch_encode_lut_t<15, 2>(k, s) : launch_encode_lut_t<15, 1>(k, s);
        if (p->view_edge == 31) return prec == 2 ? launch_encode_lut_t<31, 2>(k, s) : launch_encode_lut_t<31, 1>(k, s);
        return -2;
    }
    if (p->view_edge == 15) return prec == 2 ? launch_encode_t<15, 2, false>(k, s) : launch_encode_t<15, 1, false>(k, s);
    if (p->view_edge == 31) return prec == 2 ? launch_encode_t<31, 2, false>(k, s) : launch_encode_t<31, 1, false>(k, s);
    return -2;
}

// ---- pack: conv_w f32 [6, 3, 3, 3], lin_w f32 [32, 6 P] -> fragment images ------------------------------------------------------
template <int V, int PREC>
__global__ __launch_bounds__(256) void k_pack_encoder(const float* __restrict__ cw, const float* __restrict__ cb, const float* __restrict__ lw, uint8_t* conv_frags,
                                                      uint8_t* lin_frags, int32_t* err) {
    using G = Geo<V>;
    constexpr int O = G::O, XTP = G::XTP, P = O * O, UNITS = SSD_ENCODE_UNITS(V);
    constexpr int NCONV = 9 * 512, NLIN = UNITS * 2 * 512;
    constexpr float CS = PREC == 2 ? ENC_CSCALE : 2.f, LS = PREC == 2 ? ENC_LSCALE : 1.f;   // PREC 1: the plane value is 0.5
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (PREC == 2 && e < 6 && err) {
        // The conv ACTIVATIONS are split at the scale CS inside the encoder's loop, where a range check per value would cost ~4 % of
        // the kernel: they are bounded here instead.  A window cell lights at most one plane, so for output channel e
        // |conv| <= |b| + 255/256 * sum over the 9 taps of max_ch |w[e, ch, dy, dx]| -- the exact worst case over all observations.
        float bound = fabsf(cb[e]);
        for (int tap = 0; tap < 9; ++tap) {
            float mx = 0.f;
            for (int ch = 0; ch < 3; ++ch) mx = fmaxf(mx, fabsf(cw[(e * 3 + ch) * 9 + tap]));
            bound += mx * (255.f / 256.f);
        }
        if (!(bound * CS <= F16_MAX)) atomicOr(err, ERR_F16_RANGE);
    }
    if (e < NCONV) {
        // fragment (s, dy): row m = (o2 = m >> 3, position p = m & 7); quarter q < 3: plane q, cells 0..7; quarter 3: the tail
        // [R 8, R 9, G 8, G 9, B 8, B 9, 0, 0]; the tap index is d = cell - p
        const int sd = e >> 9, s = sd / 3, dy = sd % 3, lane = (e >> 3) & 63, j = e & 7, q = lane >> 4, m = lane & 15;
        const int oc = 2 * s + (m >> 3), pp = m & 7;
        const int ch = q < 3 ? q : j >> 1, cell = q < 3 ? j : 8 + (j & 1), d = cell - pp;
        float w = 0.f;
        if ((q < 3 || j < 6) && d >= 0 && d <= 2) w = (float)((double)cw[((oc * 3 + ch) * 3 + dy) * 3 + d] * (255.0 / 256.0) * (double)CS);
        store_term<PREC>(conv_frags + ((size_t)sd * 64 + lane) * 16 + 2 * j, w, (size_t)9 * 1024, err);
    } else if (e < NCONV + NLIN) {
        const int i = e - NCONV, mt = (i >> 9) & 1, u = i >> 10, lane = (i >> 3) & 63, j = i & 7, q = lane >> 4, m = lane & 15;
        const int s = u % 3, xtp = (u / 3) % XTP, y = u / (3 * XTP);
        const int r = 4 * q + (j & 3);                                 // row of conv result tile (j >> 2) of the pair
        const int oc = 2 * s + (r >> 3), x = 8 * (2 * xtp + (j >> 2)) + (r & 7);
        const float w = x < O ? lw[(size_t)(16 * mt + m) * (6 * P) + oc * P + y * O + x] * LS : 0.f;
        store_term<PREC>(lin_frags + (((size_t)(u * 2 + mt) * PREC) * 64 + lane) * 16 + 2 * j, w, (size_t)64 * 16, err);
    }
}

// ---- pack, class-LUT layout: conv_w / conv_b -> table f32 [3][64][6] (scaled by the conv activations' split scale), lin_w -> the
// Linear image in position-major K order (see encode_body_lut) ------------------------------------------------------------------
template <int V, int PREC>
__global__ __launch_bounds__(256) void k_pack_encoder_lut(const float* __restrict__ cw, const float* __restrict__ cb, const float* __restrict__ lw, float* table,
                                                          uint8_t* lin_frags, int32_t* err) {
    using G = Geo<V>;
    constexpr int O = G::O, R = G::R, P = O * O, KSTEPS = SSD_ENCODE_LUT_KSTEPS(V);
    constexpr int NTAB = 3 * LUT_DY_F, NLIN = KSTEPS * 2 * 512;
    constexpr float CS = PREC == 2 ? ENC_CSCALE : 1.f, LS = PREC == 2 ? ENC_LSCALE : 1.f;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (PREC == 2 && e < 6 && err) {      // range of the conv ACTIVATIONS at the scale CS (they are split inside the encoder): exact worst case, see k_pack_encoder
        float bound = fabsf(cb[e]);
        for (int tap = 0; tap < 9; ++tap) {
            float mx = 0.f;
            for (int ch = 0; ch < 3; ++ch) mx = fmaxf(mx, fabsf(cw[(e * 3 + ch) * 9 + tap]));
            bound += mx * (255.f / 256.f);
        }
        if (!(bound * CS <= F16_MAX)) atomicOr(err, ERR_F16_RANGE);
    }
    if (e < NTAB) {
        // T[dy][idx][c] = (dy == 0 ? b[c] : 0) + 255/256 * sum over dx of w[c][plane(class of cell dx)][dy][dx]; classes (SSD_OBS_CODE):
        // 2 = waste -> R (plane 0), 1 = apple -> G (plane 1), 3 = wall / agent -> B (plane 2), 0 = nothing
        const int dy = e / LUT_DY_F, idx = (e / LUT_ENTRY_F) % 64, c = e % LUT_ENTRY_F;
        double v = dy == 0 ? (double)cb[c] : 0.0;
        for (int dx = 0; dx < 3; ++dx) {
            const int cls = (idx >> (2 * dx)) & 3;
            const int plane = cls == 2 ? 0 : (cls == 1 ? 1 : (cls == 3 ? 2 : -1));
            if (plane >= 0) v += (double)cw[((c * 3 + plane) * 3 + dy) * 3 + dx] * (255.0 / 256.0);
        }
        table[e] = (float)(v * (double)CS);
    } else if (e < NTAB + NLIN) {
        const int i = e - NTAB, mt = (i >> 9) & 1, gs = i >> 10, lane = (i >> 3) & 63, j = i & 7, q = lane >> 4, m = lane & 15;
        int band = 0, base = 0;                                        // the band this K-step belongs to
#pragma unroll
        for (int k = 0; k < G::NB; ++k) { const int n = lut_band_ksteps<V>(k); if (gs >= base + n && k + 1 < G::NB) { base += n; band = k + 1; } }
        const int Rb = O - band * R < R ? O - band * R : R;
        const int p = 4 * (gs - base) + q, yl = p / O, x = p - yl * O;
        const float w = (j < 6 && yl < Rb) ? lw[(size_t)(16 * mt + m) * (6 * P) + j * P + (band * R + yl) * O + x] * LS : 0.f;
        store_term<PREC>(lin_frags + (((size_t)(gs * 2 + mt) * PREC) * 64 + lane) * 16 + 2 * j, w, (size_t)64 * 16, err);
    }
}

int launch_pack_encoder_lut(const float* cw, const float* cb, const float* lw, int V, int prec, void* table, void* lin_frags, hipStream_t s) {
    float* t = static_cast<float*>(table);
    uint8_t* l = static_cast<uint8_t*>(lin_frags);
    int32_t* err = numeric_err_word();
#define SSD_PACKL(V_, P_)                                                                                                    \
    do {                                                                                                                     \
        const int total = 3 * LUT_DY_F + SSD_ENCODE_LUT_KSTEPS(V_) * 2 * 512;                                                 \
        hipLaunchKernelGGL((k_pack_encoder_lut<V_, P_>), dim3((total + 255) / 256), dim3(256), 0, s, cw, cb, lw, t, l, err);  \
    } while (0)
    if (V == 15) { if (prec == 2) SSD_PACKL(15, 2); else SSD_PACKL(15, 1); return 0; }
    if (V == 31) { if (prec == 2) SSD_PACKL(31, 2); else SSD_PACKL(31, 1); return 0; }
#undef SSD_PACKL
    return -2;
}

int launch_pack_encoder(const float* cw, const float* cb, const float* lw, int V, int prec, void* conv_frags, void* lin_frags, hipStream_t s) {
    uint8_t *c = static_cast<uint8_t*>(conv_frags), *l = static_cast<uint8_t*>(lin_frags);
    int32_t* err = numeric_err_word();
#define SSD_PACK(V_, P_)                                                                                                     \
    do {                                                                                                                     \
        const int total = 9 * 512 + SSD_ENCODE_UNITS(V_) * 2 * 512;                                  \
        hipLaunchKernelGGL((k_pack_encoder<V_, P_>), dim3((total + 255) / 256), dim3(256), 0, s, cw, cb, lw, c, l, err);     \
    } while (0)
    if (V == 15) { if (prec == 2) SSD_PACK(15, 2); else SSD_PACK(15, 1); return 0; }
    if (V == 31) { if (prec == 2) SSD_PACK(31, 2); else SSD_PACK(31, 1); return 0; }
#undef SSD_PACK
    return -2;
}

}  // namespace ssd

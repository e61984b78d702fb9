// gymrs_step_impl.h — the hand-written CDNA4 (gfx950) per-step kernel of the batched classic-control stepper and its
// launch table.  Included by ONE translation unit per env type (gymrs_step_<env>.hip), which the build compiles in
// parallel: the table (lanes per work-item x flag sets x workgroup sizes) is minutes of compile time.
//
// One kernel template, instantiated per env type: each lane is one independent gym-rs env
//   CartPoleEnv::step     /root/reference/src/envs/classical_control/cartpole.rs:398-483
//   MountainCarEnv::step  /root/reference/src/envs/classical_control/mountain_car.rs:398-435
//   (+ reset, cartpole.rs:485-516 / mountain_car.rs:464-501, fused in for finished lanes)
//
// Shape of the work: pure streaming, memory-latency/bandwidth-bound, no contraction ->
// no MFMA.  What the measurements on MI355X dictated (profiles/, DESIGN.md "what was measured"):
//   * SoA f32 state in HBM; a work-item owns 4 consecutive lanes and moves them with one dwordx4
//     load + store per array (16 B per work-item, 1 KiB per wave instruction); actions and done flags
//     travel as one packed dword.
//   * A 2^20-lane launch is ONE generation of waves (CartPole: 512 workgroups x 8 wave64, all resident): every
//     wave loads, computes, stores once.  The kernel time is launch + memory round trip + whatever
//     VALU work and latency sit between a wave's last load and its stores.
//   * The physics of the 4 lanes of a work-item runs as ONE basic block whenever the whole wave is on
//     the common path (valid actions, angle in the polynomial's range), so the 4 independent dependency
//     chains interleave; the general per-lane code is the wave-uniform fallback.  Both give the same
//     bits.  (Packed v_pk_*_f32 math was measured to run at half rate on this part: no gain, not used.)
//   * auto-reset: about 1 lane in 22 finishes per step, so ~95 % of the 256-lane waves hold a finished
//     lane.  A wave __ballot done-mask + mbcnt ranks compact them into the wave's own LDS segment and the
//     wave evaluates ONE Philox4x32-10 block per finished lane (one pass of ~11 active lanes) instead of
//     one per lane under divergence; the fresh states return through LDS to the owning work-item, which
//     stores them with its vector store.  Everything stays inside one wavefront: no s_barrier (a
//     workgroup-level variant with one Philox pass per workgroup measured the same on CartPole and 10 %
//     slower on MountainCar: its barrier re-couples the waves), no speculative evaluation (Philox in the
//     load shadow measured no better: v_mad_u64_u32 is quarter rate and the shadow is not free).
//     Quiet waves (MountainCar, Pendulum: nearly all) skip the whole path.
//   * Episode bookkeeping: what a step has to remember is the tick each lane's open episode started at (episodes tile a
//     lane's time axis, so the sum of finished lengths is sum(ep_start) - n*epoch, evaluated when statistics are read)
//     and the number of finished episodes.  Every vector memory instruction of a wave costs 0.05-0.1 us of launch
//     time here, and a partial-line write is cheap only while its line is in the Infinity Cache: scattered ep_start
//     stores cost 0.45 us per launch, a counter's load + store 0.07.  CartPole (statistics on, no time limit) therefore
//     keeps a RESET LOG: a wave stores its VEC done-masks (32 contiguous bytes) into an 8-row ring and every 8th launch
//     folds the ring inside the kernel (step_block below).  The other variants keep the scattered stores (after a dense
//     ep_start read of the same wave they hit) and a per-wave counter slot (plain load at start, plain store at end).
#pragma once
#include "gymrs_tile.h"

namespace gymrs {

// ep = the wave-uniform bit mask `m` (an SGPR pair) selects, per work-item, `fresh` over `ep`: one v_cndmask_b32 with the
// mask as its condition operand -- a ballot result used the way the hardware uses VCC.
__device__ __forceinline__ uint32_t select_by_mask(unsigned long long m, uint32_t fresh, uint32_t ep)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(ep), "v"(fresh), "s"(m));
    return r;
}

// StepArgs::fold_step (reset-logged variants only): this launch is the kResetLogRows-th since the last fold.  Every wavefront folds ITS
// OWN column of the ring -- the done-masks of the previous kResetLogRows - 1 steps, one word per lane fetched behind the
// state loads (they were written by earlier launches), plus this step's masks from registers -- into the start ticks of
// its own lanes (one dense dwordx4 read-modify-write of ep_start per work-item per kResetLogRows steps; a mask reaches
// the lanes as the condition operand of one v_cndmask each) and into its episode counter, and zeroes the column.  No
// extra launch, no scattered store, and the ring stays 8 rows small.  Measured at 2^20 CartPole lanes: a folding launch
// costs 1.4 us more than a plain one (0.64 of it the ep_start read-modify-write), i.e. 0.17 us per step; per-step time
// 6.66 us (scattered stores, round 1) -> 6.47 us on the same box.  The folding launch is the SAME kernel taking a
// wave-uniform branch on a kernel argument, not a second instantiation: alternating two kernels (each with its own copy of
// the physics and reset code) cost MountainCar's 4 us launches up to 1.3 us each.
// vblock = the index of the workgroup-sized tile (THREADS * VEC lanes): blockIdx.x, unless the workgroup steps several tiles
// (step_kernel_body, TILES > 1).  `d` holds the tile's loads (issued, not necessarily landed).
template <class Env, int VEC, uint32_t FLAGS, int THREADS, bool FULL>
__device__ __forceinline__ void step_block_loaded(StepArgs a, const typename Env::Consts& c, ResetLds<Env, VEC, THREADS>& lds, uint32_t vblock,
                                                  TileRegs<Env, VEC, FLAGS>& d)
{
    constexpr int kVec = VEC;
    constexpr int LPB = THREADS * kVec;
    constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    constexpr bool STATS = AUTO && (FLAGS & GYMRS_TRACK_STATS);
    constexpr bool LOGGED = TileRegs<Env, VEC, FLAGS>::LOGGED; // bookkeeping through the reset log: no statistics slot to load
    const uint64_t base = (uint64_t)vblock * LPB + (uint64_t)threadIdx.x * kVec;
    // A replayed HIP graph keeps the tick on the device.  Resolved here, BEHIND the state loads: ahead of them
    // the branch makes every wave wait for the whole kernel-argument fetch before it issues its first load.
    if (a.tick_base) a.tick += *a.tick_base;
    // Episode statistics: every wavefront owns one {finished episodes, sum of returns} slot.  The old value
    // is fetched right behind the state loads (after them, so that the slot pointer does not split the
    // kernel-argument fetch in two) and the updated value leaves with the wave's last stores: the hot
    // path holds no atomic and nothing waits on the statistics.  (Launches are stream-ordered and a slot
    // has exactly one writer per launch.)
    unsigned long long old_resets = 0;
    double old_ret = 0.0, open = 0.0;
    const size_t wave_slot = (size_t)vblock * (THREADS / 64) + (threadIdx.x >> 6); // = the global wave index
    if (STATS && !LOGGED) {
        const unsigned long long* bs = a.block_stats + wave_slot * 2;
        old_resets = bs[0];
        if (!Env::kConstReward) {
            old_ret = reinterpret_cast<const double*>(bs)[1];
            open = a.wave_open[wave_slot];
        }
    }
    // Folding launch: what the fold needs is fetched right behind the state loads -- this work-item's start ticks, the
    // wave's counter, and the wave's column of the ring: lane i < 7 * VEC fetches word (i % VEC) of the row (i / VEC) + 1
    // steps back.  (Holding the 28 words in SGPRs from here on made the compiler spill 116 SGPRs: + 1 us per launch.)
    constexpr int kOlder = (int)kResetLogRows - 1;
    Vec<uint32_t, VEC> ep;
    unsigned long long older = 0;
    [[maybe_unused]] unsigned long long* older_at = nullptr;
    const bool fold = LOGGED && a.fold_step != 0; // wave-uniform (a kernel argument)
    if (fold) {
        ep = load_vec<uint32_t, kVec, false>(a.ep_start, base, a.n, FULL, 0u);
        const uint32_t lane = threadIdx.x & 63u;
        older_at = a.reset_log + (size_t)((uint32_t)(a.tick - 1 - (uint64_t)(lane / kVec)) & (kResetLogRows - 1u)) * a.reset_log_row_words +
                   wave_slot * kVec + lane % kVec;
        if (lane < (uint32_t)(kOlder * kVec)) older = *older_at;
        old_resets = a.block_stats[wave_slot * 2];
    }
    // MountainCar pays -1.0 on every step (mountain_car.rs:423; SURVEY 8a row a6 "may be elided ... but counted in
    // algorithmic bytes"): once a wave's part of `reward` holds the constant it is not rewritten until a step pays
    // something else there (an invalid action).  A per-wave flag remembers it.  CartPole under auto-reset pays a constant
    // too (cartpole.rs:455-459): its engines elide from 128 MiB per step on (StepArgs::elide_reward, a wave-uniform kernel
    // argument) -- 2^22 lanes 25.5 -> 23.5 us, 2^24 98 -> 89, 2^25 214 -> 185 (profiles/r04_cartpole_reward_elision.log) -- and
    // not below: the headline configuration moves every byte it is credited with.
    constexpr bool CAN_ELIDE = AUTO && Env::kConstReward;
    const bool elide = CAN_ELIDE && (Env::kElideConstReward || a.elide_reward != 0u);
    uint32_t clean = 0;
    if (elide) clean = a.wave_clean[wave_slot];
    // Chains only (StepArgs::xcc_table): is this workgroup index still on the XCD the chain's first launch found its residue class (mod 8) on?
    // One wavefront per workgroup asks.  The table entry was stored PLAINLY by a workgroup of that first launch, so while the deal is
    // stable it is read out of this XCD's own L2 like the state (a table written through by another XCD took ~1 us to arrive and cost
    // MountainCar's 3.1 us chain step 0.3 us: profiles/r04_xcd_check_cost.log); a workgroup that finds itself elsewhere reads whatever memory
    // holds -- an older chain's tag -- and reports.  Both reads sit HERE, behind every load of the step: the wave waits for memory anyway
    // (s_getreg is a slow scalar instruction; at the end of the kernel it kept every wave resident 0.25 us longer).
#ifndef GYMRS_EXP_XCC_PARTS // (developer builds, A/B runs: 1 the table fetch, 2 the s_getreg, 4 the compare / record at the end; 0 = no check)
#define GYMRS_EXP_XCC_PARTS 7
#endif
    uint32_t xcc_want = 0, xcc_id = 0;
    const bool xcc_asks = a.xcc_check != 0u && (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64u && vblock == blockIdx.x * (uint32_t)kStepTiles;
#if GYMRS_EXP_XCC_PARTS & 1
    if (xcc_asks && a.xcc_check == 1u) xcc_want = a.xcc_table[(blockIdx.x & 7u) * kXccTableStride];
#endif
#if GYMRS_EXP_XCC_PARTS & 2
    if (xcc_asks) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        xcc_id = (a.xcc_seq << 8) | ((xcc_id & 0xfu) + 1u); // tagged with the chain's number: an entry of an earlier chain never matches
    }
#endif
    GYMRS_STAMP(1);
    StepOut<VEC> out;
    advance_tile<Env, VEC, FLAGS, FULL, false, THREADS>(a, c, base, d, lds, old_resets, old_ret, open, out, vblock);
    store_tile<Env, VEC, FLAGS, FULL>(a, base, d, out, elide && clean != 0 && out.reward_is_const);
    if (fold) {
        const uint32_t lane = threadIdx.x & 63u;
        // finished episodes of the wave in the ring's 8 steps: popcounts of the older words (one per lane) and of this step's masks
        uint32_t finished = wave_sum_u32((uint32_t)__popcll(older));
#pragma unroll
        for (int k = 0; k < kVec; ++k) finished += (uint32_t)__popcll(out.masks[k]);
        if (finished != 0) { // quiet waves (MountainCar: most) leave everything as it is
#pragma unroll
            for (int s = kOlder - 1; s >= 0; --s) { // oldest first, so that the most recent re-arm of a lane wins
                const uint32_t fresh = (uint32_t)(a.tick - (uint64_t)s); // the episode began at the tick after that step
#pragma unroll
                for (int k = 0; k < kVec; ++k) {
                    const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(older >> 32), s * kVec + k) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)older, s * kVec + k);
                    ep.v[k] = select_by_mask(m, fresh, ep.v[k]);
                }
            }
            const uint32_t fresh_now = (uint32_t)(a.tick + 1);
#pragma unroll
            for (int k = 0; k < kVec; ++k) ep.v[k] = select_by_mask(out.masks[k], fresh_now, ep.v[k]);
            store_vec<uint32_t, kVec, false>(a.ep_start, base, a.n, FULL, ep);
            if (older != 0) *older_at = 0; // zero the column (only lanes < 7 * VEC hold a word)
            if (lane == 0) a.block_stats[wave_slot * 2] = old_resets + finished;
        }
    }
    if (elide && (clean != 0) != out.reward_is_const && (threadIdx.x & 63u) == 0) a.wave_clean[wave_slot] = out.reward_is_const ? 1u : 0u;
    if (STATS && !Env::kConstReward && (threadIdx.x & 63u) == 0) a.wave_open[wave_slot] = open;
#if GYMRS_EXP_XCC_PARTS & 4
    if (xcc_asks) { // (see above) the first launch of a chain records, every later one compares
        if (a.xcc_check == 2u) {
            if (blockIdx.x < 8u && threadIdx.x == 0) a.xcc_table[blockIdx.x * kXccTableStride] = xcc_id; // (plain: stays in this XCD's L2)
        } else if (xcc_want != xcc_id && threadIdx.x == 0) {
            a.err_seen[1] = blockIdx.x + 1u;
        }
    }
#endif
    GYMRS_STAMP(6);
}

template <class Env, int VEC, uint32_t FLAGS, int THREADS, bool FULL>
__device__ __forceinline__ void step_block(const StepArgs& a, const typename Env::Consts& c, ResetLds<Env, VEC, THREADS>& lds, uint32_t vblock)
{
    GYMRS_STAMP(0);
    TileRegs<Env, VEC, FLAGS> d;
    load_tile<Env, VEC, FLAGS, FULL>(a, (uint64_t)vblock * (THREADS * VEC) + (uint64_t)threadIdx.x * VEC, d);
    step_block_loaded<Env, VEC, FLAGS, THREADS, FULL>(a, c, lds, vblock, d);
}

// VEC lanes per work-item: 4 (4096 waves for 2^20 lanes, 4 waves per SIMD) or 8 (2 waves per SIMD; a tuning knob: measured
// 9.9 vs 6.8 us; 16 lanes, 28 us, were removed).  More lanes per wave = fewer waves = fewer per-wave fixed costs (address set-up, the
// auto-reset Philox pass, which costs the same whether 11 or 45 of its 64 lanes are active) at the price of
// registers; waves_per_eu lets the allocator use them instead of spilling to chase occupancy.
// The pointers and the lane count the first instructions of a wave need are separate scalar kernel parameters
// placed first, so that kernel-argument preloading (-mllvm -amdgpu-kernarg-preload-count, see build.py) can
// deliver them in SGPRs at wave launch instead of behind an s_load round trip; the rest travels in StepArgs.
// THREADS work-items per workgroup: 256, or Env::kThreads (CartPole: 512) for launches big enough to still put two
// workgroups on every CU -- small batches want many small workgroups (16384 lanes: 3.0 vs 3.6 us).
// Where the kernel arguments come from (round 6; Env::kLoadsAheadOfArguments: CartPole, Pendulum).  The six leading scalars arrive PRELOADED in SGPRs; everything else
// (StepArgs, Consts) lies in the kernel-argument segment behind a scalar-load round trip -- two of them, in fact, and 16 v_writelane of spilled SGPRs, because the
// compiler fetches every used by-value argument at the top of the kernel and holds all of it live: a wave issued its first state load only after both
// (profiles/r06_wave_phase_trace_2p21.log: "issue loads" 650-1170 cycles per wave).  For such an env the body never names `rest` / `c`: it issues the tile's loads
// from the preloaded scalars alone and THEN reads the segment through its pointer (same bytes, same layout: StepKernArgs), so that the argument fetch runs in the
// shadow of the state loads.  profiles/r06_loads_before_arguments.log: CartPole 2^20 lanes 6.39-6.47 -> 6.09-6.31 us in eight run pairs of eight on four boxes,
// other sizes within +-1.5 %; Pendulum (one early wait, no spill) 2^20 lanes 5.9 -> 5.4 us, 2^22 -2 .. -5 % on three boxes and +6 % once.  MountainCar's kernel
// already had its loads ahead of the wait and loses 1-4 % from 2^22 lanes on in this structure (EITHER order): it keeps the by-value form, byte-identical to round 5's.
template <class Env>
struct KernArgView {
    using Consts = typename Env::Consts;
    static __device__ __forceinline__ void fetch(StepArgs& a, Consts& c)
    {
        // (the segment lies in the constant address space; the cast to a plain pointer is folded back by address-space inference: scalar loads, and the pointer
        // members stay what kernel-argument pointers are -- global.  Copied word by word instead, they would come out as flat pointers: 50 flat_* instructions.)
        const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        __builtin_memcpy(&a, kp + offsetof(StepKernArgs<Consts>, rest), sizeof(StepArgs));
        __builtin_memcpy(&c, kp + offsetof(StepKernArgs<Consts>, c), sizeof(Consts));
    }
};

template <class Env, int VEC, uint32_t FLAGS, int THREADS>
__device__ __forceinline__ void step_kernel_body(float* s0, float* s1, float* s2, float* s3, const void* action, uint64_t n_fast, const StepArgs& rest,
                                                 const typename Env::Consts& c_by_value)
{
    constexpr int LPB = THREADS * VEC;
    __shared__ ResetLds<Env, VEC, THREADS> lds;
    // wave-uniform: every wavefront whose 64 * VEC lanes all exist runs the unguarded body; only the one that
    // straddles n (and the empty ones behind it) takes the guarded per-lane code.  n_fast (a preloaded scalar
    // argument) is n -- or 0 when the caller's action buffer is not aligned for the vector load, which sends
    // every wavefront through the guarded code (per-lane action loads); the real n travels in StepArgs.
    if constexpr (kStepTiles == 1 && Env::kLoadsAheadOfArguments) {
        StepArgs a;
        typename Env::Consts c;
        if ((uint64_t)blockIdx.x * LPB + (uint64_t)((threadIdx.x >> 6) + 1) * (64 * VEC) <= n_fast) {
#ifdef GYMRS_TRACE_TIMES
            const unsigned long long t_start = __builtin_amdgcn_s_memtime(); // (stamp 0 is stored below: the trace buffer's pointer is an argument too)
#endif
            using R = TileRegs<Env, VEC, FLAGS>;
            using Action = typename Env::Action;
            const uint64_t base = (uint64_t)blockIdx.x * LPB + (uint64_t)threadIdx.x * VEC;
            float* const sp[4] = {s0, s1, s2, s3};
            TileRegs<Env, VEC, FLAGS> d;
#pragma unroll
            for (int j = 0; j < Env::kState; ++j) d.st[j] = load_vec<float, VEC, R::NT_SL>(sp[j], base, n_fast, true, 0.0f);
            d.act = load_vec<Action, VEC, R::NT_A>(static_cast<const Action*>(action), base, n_fast, true, Action(0));
            __builtin_amdgcn_sched_barrier(0); // nothing of the argument fetch below moves ahead of these loads
            KernArgView<Env>::fetch(a, c);
            a.s[0] = s0;
            a.s[1] = s1;
            a.s[2] = s2;
            a.s[3] = s3;
            a.action = action;
#ifdef GYMRS_TRACE_TIMES
            if (a.trace && (threadIdx.x & 63u) == 0) a.trace[((size_t)blockIdx.x * a.trace_wpb + (threadIdx.x >> 6)) * 8 + 0] = t_start;
#endif
            // (the arrays only some flag sets read: their pointers travel in StepArgs)
            if (Env::kHasBeyond && !R::AUTO) d.beyond = load_vec<uint8_t, VEC, R::NT_SL>(a.beyond, base, a.n, true, uint8_t(0));
            if (R::TLIM && !Env::kNeverTerminates) d.ep_start = load_vec<uint32_t, VEC, false>(a.ep_start, base, a.n, true, 0u);
            step_block_loaded<Env, VEC, FLAGS, THREADS, true>(a, c, lds, blockIdx.x, d);
        } else {
            KernArgView<Env>::fetch(a, c);
            a.s[0] = s0;
            a.s[1] = s1;
            a.s[2] = s2;
            a.s[3] = s3;
            a.action = action;
            step_block<Env, VEC, FLAGS, THREADS, false>(a, c, lds, blockIdx.x);
        }
    } else {
        const typename Env::Consts& c = c_by_value;
        StepArgs a = rest;
        a.s[0] = s0;
        a.s[1] = s1;
        a.s[2] = s2;
        a.s[3] = s3;
        a.action = action;
        if constexpr (kStepTiles == 1) {
            if ((uint64_t)blockIdx.x * LPB + (uint64_t)((threadIdx.x >> 6) + 1) * (64 * VEC) <= n_fast)
                step_block<Env, VEC, FLAGS, THREADS, true>(a, c, lds, blockIdx.x);
            else
                step_block<Env, VEC, FLAGS, THREADS, false>(a, c, lds, blockIdx.x);
        } else {
            // (developer builds, GYMRS_EXP_TILES) a workgroup steps kStepTiles consecutive tiles, the loads of tile j + 1 issued before the
            // arithmetic of tile j: more bytes in flight per resident wave where a launch is several generations of waves
            const uint32_t vb0 = blockIdx.x * (uint32_t)kStepTiles;
            if ((uint64_t)(vb0 + kStepTiles - 1) * LPB + (uint64_t)((threadIdx.x >> 6) + 1) * (64 * VEC) <= n_fast) {
                TileRegs<Env, VEC, FLAGS> d[2];
                load_tile<Env, VEC, FLAGS, true>(a, (uint64_t)vb0 * LPB + (uint64_t)threadIdx.x * VEC, d[0]);
#pragma unroll
                for (int j = 0; j < kStepTiles; ++j) {
                    if (j + 1 < kStepTiles) load_tile<Env, VEC, FLAGS, true>(a, (uint64_t)(vb0 + j + 1) * LPB + (uint64_t)threadIdx.x * VEC, d[(j + 1) & 1]);
                    step_block_loaded<Env, VEC, FLAGS, THREADS, true>(a, c, lds, vb0 + j, d[j & 1]);
                }
            } else {
                for (int j = 0; j < kStepTiles; ++j)
                    if ((uint64_t)(vb0 + j) * LPB < rest.n) step_block<Env, VEC, FLAGS, THREADS, false>(a, c, lds, vb0 + j);
            }
        }
    }
}

#ifndef GYMRS_EXP_WAVES // (developer builds: the occupancy the register allocator aims for)
#define GYMRS_EXP_WAVES(VEC_) (16 / (VEC_) < 1 ? 1 : 16 / (VEC_))
#endif
#ifndef GYMRS_EXP_WAVES_MIN
#define GYMRS_EXP_WAVES_MIN 1
#endif
#define GYMRS_STEP_KERNEL_ATTRS(THREADS_, VEC_) \
    __global__ __launch_bounds__(THREADS_) __attribute__((amdgpu_waves_per_eu(GYMRS_EXP_WAVES_MIN, GYMRS_EXP_WAVES(VEC_))))

template <class Env, int VEC, uint32_t FLAGS, int THREADS>
GYMRS_STEP_KERNEL_ATTRS(THREADS, VEC) void step_kernel(float* s0, float* s1, float* s2, float* s3, const void* action, uint64_t n_fast,
                                                       const StepArgs rest, const typename Env::Consts c)
{
    step_kernel_body<Env, VEC, FLAGS, THREADS>(s0, s1, s2, s3, action, n_fast, rest, c);
}

#ifndef GYMRS_EXP_BLOCK
static_assert(CartPoleT::kThreads == kCartPoleThreads && MountainCarT::kThreads == kBlock && PendulumT::kThreads == kBlock,
              "step_threads_of (gymrs_kernels.h) must pick what launch_one picks");
#endif

// ---------------------------------------------------------------------------------------------
// launch tables
template <class Env, int VEC, uint32_t FLAGS>
static hipError_t launch_one(const StepArgs& a, const void* consts, hipStream_t stream)
{
    launch_begin(); // (gymrs_kernels.h: the thread's last-error word may hold somebody else's error)
    if constexpr (Env::kThreads != kBlock) {
        if (step_uses_big_groups(a.n, Env::kThreads, VEC)) { // >= 2 big workgroups per CU, at most two generations of waves
            hipLaunchKernelGGL((step_kernel<Env, VEC, FLAGS, Env::kThreads>), dim3(step_grid(a.n, VEC, Env::kThreads * kStepTiles)), dim3(Env::kThreads), 0,
                               stream, a.s[0], a.s[1], a.s[2], a.s[3], a.action, a.n_fast, a, *static_cast<const typename Env::Consts*>(consts));
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((step_kernel<Env, VEC, FLAGS, kBlock>), dim3(step_grid(a.n, VEC, kBlock * kStepTiles)), dim3(kBlock), 0, stream, a.s[0], a.s[1],
                       a.s[2], a.s[3], a.action, a.n_fast, a, *static_cast<const typename Env::Consts*>(consts));
    return hipGetLastError();
}

template <class Env, int VEC, uint32_t NTBIT>
static hipError_t launch_flags_nt(uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    constexpr uint32_t A = GYMRS_AUTO_RESET, S = GYMRS_TRACK_STATS, T = GYMRS_TIME_LIMIT;
    if (!(flags & A)) flags &= ~S; // statistics need auto-reset
    switch (flags & (A | S | T)) {
#ifdef GYMRS_DEV_MINIMAL // developer builds (tools/devbuild.py): only the headline flag sets, seconds instead of minutes
    case A | S: return launch_one<Env, VEC, A | S | NTBIT>(a, consts, stream);
    case A | S | T: return launch_one<Env, VEC, A | S | T | NTBIT>(a, consts, stream);
    default: return hipErrorInvalidValue;
#else
    case 0: return launch_one<Env, VEC, 0 | NTBIT>(a, consts, stream);
    case A: return launch_one<Env, VEC, A | NTBIT>(a, consts, stream);
    case A | S: return launch_one<Env, VEC, A | S | NTBIT>(a, consts, stream);
    case T: return launch_one<Env, VEC, T | NTBIT>(a, consts, stream);
    case A | T: return launch_one<Env, VEC, A | T | NTBIT>(a, consts, stream);
    case A | S | T: return launch_one<Env, VEC, A | S | T | NTBIT>(a, consts, stream);
    default: return hipErrorInvalidValue;
#endif
    }
}

// hint variants of a HIP launch: every access (small batches and far beyond the caches), only the stores nobody reads again
// (kFlagNtOut: in between, profiles/r04_hints_by_size.log), none
template <class Env, int VEC>
static hipError_t launch_flags(uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    if (flags & kFlagNonTemporal) return launch_flags_nt<Env, VEC, kFlagNonTemporal>(flags, a, consts, stream);
    if (flags & kFlagNtOut) return launch_flags_nt<Env, VEC, kFlagNtOut>(flags, a, consts, stream);
    return launch_flags_nt<Env, VEC, 0u>(flags, a, consts, stream);
}

template <class Env>
static hipError_t launch_vec(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    switch (vec) {
    case 4: return launch_flags<Env, 4>(flags, a, consts, stream);
#ifndef GYMRS_DEV_MINIMAL
    case 8: return launch_flags<Env, 8>(flags, a, consts, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}

} // namespace gymrs

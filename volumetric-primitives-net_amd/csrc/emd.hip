// emd.hip — approximate Earth Mover's Distance by the auction algorithm, forward and backward (gfx950).
//
// Replaces the reference's only in-tree native code, modules/loss/emd/emd_cuda.cu (9 kernels, host loop
// emd_cuda_forward :228-282 launching 7 of them per iteration, `iters` = 50 in training, train.py:193) behind
// EarthMoverDistanceLoss (modules/loss/emd/emd_module.py:29-78).  Same algorithm, round for round:
//   every unassigned point i bids for the target j maximising  v = 3 - |x1_i - x2_j| - price_j  (Bid :95-179),
//   the bid increment is  best - second_best + eps;  per target the highest increment wins (GetMax :181-194),
//   the winner takes the target, evicts its previous owner and raises the price (Assign :196-215);
//   the last iteration assigns every remaining point to its bid;  dist = squared distance to the assigned target.
//
// MI355X design: ONE persistent workgroup (1024 lanes = 16 waves) per sample runs all iterations in a single
// launch (the reference needs 7 x iters launches).  Targets and prices stream through LDS tiles (SoA, conflict
// free); each wave takes one bidding point at a time, its 64 lanes scan the targets and a shuffle reduce yields
// (best, lowest best index, second best).  All tie rules are deterministic (lowest index), unlike the reference's
// last-writer-wins races (GetMax :189-191, `last` Assign), so the result is reproducible and equal to the CPU
// oracle (oracle/vpn_oracle.py::emd_auction) bit for bit.  Parity with the CUDA extension itself is unpinned: it
// cannot be built here (no nvcc) and ships no stored answers (its only check is test_emd, emd_module.py:81-95).
#include "vpn_common.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace vpn {

constexpr int EMD_THREADS = 1024;
constexpr int EMD_WAVES = EMD_THREADS / 64;
constexpr int EMD_TILE = 4096;           // targets per LDS tile: 4 planes x 16 KB
constexpr int EMD_UNROLL = 4;            // targets per lane per step of the scan; tiles are padded to 64 * EMD_UNROLL
constexpr int EMD_WS_PLANES = 10;        // 4-byte words of workspace per point (8 used by the streaming kernel, 10 by the replicated one)
constexpr int EMD_MAX_GROUP = 16;        // workgroups cooperating on one sample

// one bidder's running result over a set of targets
struct Bid3 { float best, better; int idx; };

// fold the result over a disjoint target set into `a`: largest value (lowest index among equals) and the second
// largest counting duplicates — what one serial scan in index order gives (emd_cuda.cu:144-151, :163-173)
__device__ inline void emd_merge(Bid3& a, float ob, float obt, int oi) {
    const bool take = oi >= 0 && (a.idx < 0 || ob > a.best || (ob == a.best && oi < a.idx));
    a.better = take ? fmaxf(obt, a.best) : fmaxf(a.better, ob);
    a.best = take ? ob : a.best;
    a.idx = take ? oi : a.idx;
}

// Correctly rounded sqrt for 0 <= s < 2^100 without the compiler's denormal rescaling: v_sqrt_f32 (1 ulp) and the
// two-residual fix-up.  For s < 2^-96 (where the rescaling would matter) the result is only required to be below
// 2^-24, because the caller subtracts it from 3.0f.
__device__ inline float emd_sqrt(float s) {
    const float r = __builtin_amdgcn_sqrtf(s);
    const float lo = __int_as_float(__float_as_int(r) - 1), hi = __int_as_float(__float_as_int(r) + 1);
    const float elo = __builtin_fmaf(-lo, r, s), ehi = __builtin_fmaf(-hi, r, s);
    float q = (0.0f >= elo) ? lo : r;
    q = (0.0f < ehi) ? hi : q;
    return q;
}

// State shared by the workgroups of a sample (assign, assign_inv, price) is read and written past the per-CU L1
// (agent-scope relaxed accesses: sc1 loads / write-through stores), so the barrier below needs no cache-wide
// write-back or invalidate — those cost ~70 us per barrier here, the accesses cost nothing measurable.
template <typename T>
__device__ inline T emd_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ inline void emd_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Barrier over the G workgroups of one sample.  Protocol (the hand-off form MI355X_MICROARCH.md lists as valid
// without cache-wide fences): every byte of shared state is stored and loaded with agent-scope (sc1) accesses
// (emd_st / emd_ld); every wave waits for its own stores and atomics to be acknowledged (vmcnt counts stores on gfx9)
// BEFORE the workgroup barrier; only then one lane adds to the group's counter and polls it with sc1 loads; the other
// waves load shared state only after the second workgroup barrier.
// Co-residency of the G workgroups is required while they spin: the host launches this kernel COOPERATIVELY when
// G > 1 (the runtime rejects a grid that cannot be resident) and falls back to G = 1 otherwise.  As a last resort
// the spin is bounded: after ~1 s (other work holding the CUs, a partitioned device) the workgroup gives up, flags
// the sample and the kernel writes NaN distances for it instead of hanging the GPU.
constexpr unsigned EMD_SPIN_LIMIT = 1u << 24;          // x s_sleep(1) = 64 cycles each: ~0.5 s at 2.1 GHz

__device__ inline bool emd_group_sync(unsigned* counter, unsigned& passed, int G, int* gave_up) {
    if (G == 1) { __syncthreads(); return true; }
    __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    passed += G;
    // -DEMD_STRICT_ORDER: release / acquire on the counter, i.e. the form the HIP memory model asks for (the compiler
    // adds an L2 write-back before the add and an invalidate after the load: ~70 us per barrier, G > 1 then loses to
    // G = 1).  The default relies on the hardware argument above: all shared state is sc1-accessed and acknowledged.
#if defined(EMD_STRICT_ORDER) || defined(VPN_STRICT_ORDER)
    constexpr int EMD_ADD_ORDER = __ATOMIC_RELEASE, EMD_POLL_ORDER = __ATOMIC_ACQUIRE;
#else
    constexpr int EMD_ADD_ORDER = __ATOMIC_RELAXED, EMD_POLL_ORDER = __ATOMIC_RELAXED;
#endif
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, EMD_ADD_ORDER, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, EMD_POLL_ORDER, __HIP_MEMORY_SCOPE_AGENT) < passed) {
            __builtin_amdgcn_s_sleep(1);
            // the give-up flag of the group is polled every 256th spin only: a second L2 round trip per spin would
            // lengthen every barrier of every round
            if (++spins > EMD_SPIN_LIMIT ||
                ((spins & 255u) == 0 && __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tell the others
                *gave_up = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *gave_up == 0;
}

// grid: (ceil(B/8) * 8 * G) workgroups; the G workgroups of a sample sit on one XCD (workgroups are dealt to the
// 8 XCDs round-robin), so its state stays in that XCD's L2.
__global__ __launch_bounds__(EMD_THREADS) void emd_auction_kernel(const float* __restrict__ xyz1,
                                                                  const float* __restrict__ xyz2, int B, int n,
                                                                  int G, float eps, int iters,
                                                                  float* __restrict__ dist, int32_t* assignment,
                                                                  float* wsf, unsigned* counters) {
    __shared__ __attribute__((aligned(16))) float tx[EMD_TILE], ty[EMD_TILE], tz[EMD_TILE], tp[EMD_TILE];
    __shared__ int wcount[EMD_WAVES];
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / G) * 8 + xcd, g = q % G;
    if (b >= B) return;                                         // padding workgroups of a ragged batch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gtid = g * EMD_THREADS + tid, gthreads = G * EMD_THREADS;
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * n * 3;
    int32_t* assign = assignment + (size_t)b * n;
    float* base = wsf + (size_t)b * EMD_WS_PLANES * n;
    unsigned long long* top = reinterpret_cast<unsigned long long*>(base);   // per target: (increment, ~bidder) max
    float* price = base + 2 * n;                                // per target
    int* assign_inv = reinterpret_cast<int*>(base + 3 * n);     // per target: current owner
    float* inc = base + 4 * n;                                  // per point: running best, then the bid increment
    float* second = base + 5 * n;                               // per point: running second best
    int* bid = reinterpret_cast<int*>(base + 6 * n);            // per point: target it bids for
    int* ulist = reinterpret_cast<int*>(base + 7 * n);          // unassigned points, ascending
    unsigned* counter = counters + 2 * b;                       // [arrivals, gave-up flag] of this sample's group
    unsigned passed = 0;
    bool ok = true;

    for (int j = gtid; j < n; j += gthreads) {                  // emd_module.py:44-50 initial state
        emd_st(assign + j, -1); emd_st(assign_inv + j, -1); emd_st(price + j, 0.0f); emd_st(top + j, 0ull);
    }
    ok = emd_group_sync(counter, passed, G, &gave_up);

    const int per = (n + EMD_THREADS - 1) / EMD_THREADS, j0 = min(n, tid * per), j1 = min(n, j0 + per);
    for (int it = 0; ok && it < iters; ++it) {
        const bool last = it == iters - 1;
        // ---- unassigned points in ascending order (calc_unass_cnt .. calc_unass_idx :30-93; the reference's order
        //      depends on atomics, the bids do not depend on the order).  Every workgroup of the sample builds the
        //      same list (same values to the same addresses).
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += emd_ld(assign + j) == -1;
        int scan = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(scan, o, 64); if (lane >= o) scan += t2; }
        __syncthreads();                                        // previous round's readers of wcount are done
        if (lane == 63) wcount[wave] = scan;
        __syncthreads();
        int before = 0, U = 0;
#pragma unroll
        for (int w = 0; w < EMD_WAVES; ++w) { const int c = wcount[w]; before += w < wave ? c : 0; U += c; }
        if (U == 0) break;                                      // uniform over the group: later iterations are no-ops
        int pos = before + scan - cnt;
        for (int j = j0; j < j1; ++j) if (emd_ld(assign + j) == -1) ulist[pos++] = j;
        __syncthreads();

        // ---- Bid (:95-179): wave w of workgroup g takes bidders g*16+w, +16G, ...; its 64 lanes scan the targets
        for (int t0 = 0; t0 < n; t0 += EMD_TILE) {
            const int tn = min(EMD_TILE, n - t0);
            const int tpad = (tn + 64 * EMD_UNROLL - 1) / (64 * EMD_UNROLL) * (64 * EMD_UNROLL);
            const bool final_tile = t0 + EMD_TILE >= n;
            if (t0 > 0) __syncthreads();
            const bool coords = n > EMD_TILE || it == 0;        // a one-tile problem keeps its coordinates in LDS: only
            for (int j = tid; j < tpad; j += EMD_THREADS) {     // the prices change from round to round
                if (j < tn) {
                    if (coords) {
                        const float* c = p2 + (size_t)(t0 + j) * 3;
                        tx[j] = c[0]; ty[j] = c[1]; tz[j] = c[2];
                    }
                    tp[j] = emd_ld(price + t0 + j);
                } else {                                        // padding never wins: value = -inf
                    tx[j] = 0.0f; ty[j] = 0.0f; tz[j] = 0.0f; tp[j] = __builtin_inff();
                }
            }
            __syncthreads();
            for (int u = g * EMD_WAVES + wave; u < U; u += G * EMD_WAVES) {
                const int i = ulist[u];
                const float x1 = p1[i * 3], y1 = p1[i * 3 + 1], z1 = p1[i * 3 + 2];
                Bid3 r{-1e9f, -1e9f, -1};                       // :116
                for (int k = lane; k < tpad; k += 64 * EMD_UNROLL) {
#pragma unroll
                    for (int e = 0; e < EMD_UNROLL; ++e) {
                        const int kk = k + 64 * e;
                        const float dx = tx[kk] - x1, dy = ty[kk] - y1, dz = tz[kk] - z1;              // :139-141
                        const float d = (3.0f - emd_sqrt(((dx * dx) + (dy * dy)) + (dz * dz))) - tp[kk];   // :143
                        r.idx = d > r.best ? t0 + kk : r.idx;                                          // :144-151
                        r.better = __builtin_amdgcn_fmed3f(r.best, d, r.better);
                        r.best = fmaxf(r.best, d);
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(r.best, o, 64), obt = __shfl_xor(r.better, o, 64);
                    const int oi = __shfl_xor(r.idx, o, 64);
                    emd_merge(r, ob, obt, oi);
                }
                if (lane == 0) {
                    if (t0 > 0) {                               // earlier tiles hold lower indices: they are `a`
                        Bid3 a{inc[i], second[i], bid[i]};
                        emd_merge(a, r.best, r.better, r.idx);
                        r = a;
                    }
                    if (final_tile) {
                        const float v = (r.best - r.better) + eps;                          // :175-176
                        bid[i] = r.idx; inc[i] = v;
                        // :177 + GetMax :181-194 in one step: the largest increment holds the target, the lowest
                        // bidder among equal ones (v >= 0: its bit pattern orders like the value)
                        const unsigned long long key =
                            ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)(0x7fffffff - i);
                        __hip_atomic_fetch_max(top + r.idx, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        bid[i] = r.idx; inc[i] = r.best; second[i] = r.better;
                    }
                }
            }
        }
        if (!(ok = emd_group_sync(counter, passed, G, &gave_up))) break;

        // ---- Assign (:196-215): each workgroup settles the bidders it bid for (bid / inc stay CU-local)
        for (int v = tid; v < (U + G * EMD_WAVES - 1) / (G * EMD_WAVES) * EMD_WAVES; v += EMD_THREADS) {
            const int u = (v / EMD_WAVES) * (G * EMD_WAVES) + g * EMD_WAVES + (v % EMD_WAVES);
            if (u >= U) continue;
            const int i = ulist[u], t = bid[i];
            if (last) { emd_st(assign + i, t); continue; }      // prices and owners are not read again
            const unsigned long long key = emd_ld(top + t);     // the three loads depend only on t: issued together,
            const int prev = emd_ld(assign_inv + t);            // one L2 round trip instead of a chain of three
            const float pt = emd_ld(price + t);
            if (0x7fffffff - (int)(unsigned)key != i) continue;
            if (prev != -1) emd_st(assign + prev, -1);
            emd_st(assign_inv + t, i);
            emd_st(assign + i, t);
            emd_st(price + t, pt + inc[i]);
            emd_st(top + t, 0ull);                              // :212
        }
        ok = emd_group_sync(counter, passed, G, &gave_up);
    }

    if (!ok) {                                                  // the group barrier timed out: no result for this sample
        for (int j = gtid; j < n; j += gthreads) { dist[(size_t)b * n + j] = __builtin_nanf(""); assign[j] = -1; }
        return;
    }
    for (int j = gtid; j < n; j += gthreads) {                  // CalcDist :217-226
        const int t = emd_ld(assign + j);
        if (t < 0 || t >= n) { dist[(size_t)b * n + j] = __builtin_nanf(""); continue; }    // unassigned: no target to measure to
        const float dx = p1[j * 3] - p2[t * 3], dy = p1[j * 3 + 1] - p2[t * 3 + 1], dz = p1[j * 3 + 2] - p2[t * 3 + 2];
        dist[(size_t)b * n + j] = ((dx * dx) + (dy * dy)) + (dz * dz);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One-tile problems (n <= EMD_TILE, i.e. every training call: train.py:193 has n = 2048): REPLICATED STATE.
// After the first three rounds an auction round has ~100 bidders per sample (2048 points, eps 0.005): it is a chain of
// latencies, not work -- in the kernel above: rebuild the list from `assign` in memory, re-stage the prices, scan,
// group barrier, Assign through memory, group barrier.  Here every workgroup of the sample keeps its OWN copy of the
// auction state in LDS (assign, assign_inv, and the prices inside the target tile, which is staged once for the whole
// auction) and all copies evolve identically:
//   list of unassigned points from the LDS copy -> Bid scan against the LDS tile (each workgroup its share of the
//   bidders) -> each bidder publishes (target, increment) and posts the 64-bit atomic max on the target -> ONE group
//   barrier -> EVERY workgroup applies the Assign step of ALL bidders to its own copy (one lane per bidder; winners of
//   different targets touch disjoint state, so the parallel update is deterministic).
// One barrier per round instead of two, no price re-staging, no state round trips through memory.  What crosses
// workgroups: bid / increment per point (double-buffered by round parity: a loser bids again in the next round while a
// slower workgroup may still be reading this round's entry) and the per-target atomic max (three buffers: the one
// round r wrote is read in the interval after barrier r and zeroed, by slices, in the interval after barrier r + 1).
// Same arithmetic and tie rules as emd_auction_kernel: bit-equal to the oracle for every group size.
__global__ __launch_bounds__(EMD_THREADS) void emd_auction_local_kernel(const float* __restrict__ xyz1,
                                                                        const float* __restrict__ xyz2, int B, int n,
                                                                        int npad, int G, float eps, int iters,
                                                                        float* __restrict__ dist, int32_t* assignment,
                                                                        float* wsf, unsigned* counters) {
    extern __shared__ __attribute__((aligned(16))) float emd_lds[];      // 4 planes of npad floats + 3 arrays of npad ints
    float* tx = emd_lds; float* ty = tx + npad; float* tz = ty + npad; float* tp = tz + npad;
    int* assign_l = reinterpret_cast<int*>(tp + npad);
    int* inv_l = assign_l + npad;
    int* ulist = inv_l + npad;
    __shared__ int wcount[EMD_WAVES];
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / G) * 8 + xcd, g = q % G;
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * n * 3;
    float* base = wsf + (size_t)b * EMD_WS_PLANES * n;
    unsigned long long* top = reinterpret_cast<unsigned long long*>(base);     // 3 buffers of n keys: (increment, ~bidder) max
    int* gbid = reinterpret_cast<int*>(base + 6 * n);                          // 2 buffers of n: target a point bids for
    float* ginc = base + 8 * n;                                                // 2 buffers of n: its bid increment
    unsigned* counter = counters + 2 * b;
    unsigned passed = 0;
    bool ok = true;

    for (int j = tid; j < npad; j += EMD_THREADS) {             // the whole auction's tile and state (emd_module.py:44-50)
        if (j < n) { const float* c = p2 + (size_t)j * 3; tx[j] = c[0]; ty[j] = c[1]; tz[j] = c[2]; tp[j] = 0.0f; }
        else { tx[j] = 0.0f; ty[j] = 0.0f; tz[j] = 0.0f; tp[j] = __builtin_inff(); }      // padding never wins
        assign_l[j] = -1; inv_l[j] = -1;
    }
    for (int j = g * EMD_THREADS + tid; j < 3 * n; j += G * EMD_THREADS) emd_st(top + j, 0ull);
    ok = emd_group_sync(counter, passed, G, &gave_up);

    const int per = (n + EMD_THREADS - 1) / EMD_THREADS, j0 = min(n, tid * per), j1 = min(n, j0 + per);
    for (int it = 0; ok && it < iters; ++it) {
        const bool last = it == iters - 1;
        unsigned long long* top_w = top + (size_t)(it % 3) * n;              // this round's keys
        unsigned long long* top_z = top + (size_t)((it + 1) % 3) * n;        // read two rounds ago: zeroed now
        int* bid_w = gbid + (size_t)(it & 1) * n;
        float* inc_w = ginc + (size_t)(it & 1) * n;
        // ---- unassigned points in ascending order, from this workgroup's own copy
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += assign_l[j] == -1;
        int scan = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(scan, o, 64); if (lane >= o) scan += t2; }
        if (lane == 63) wcount[wave] = scan;
        __syncthreads();
        int before = 0, U = 0;
#pragma unroll
        for (int w = 0; w < EMD_WAVES; ++w) { const int c = wcount[w]; before += w < wave ? c : 0; U += c; }
        if (U == 0) break;                                      // every copy agrees
        int pos = before + scan - cnt;
        for (int j = j0; j < j1; ++j) if (assign_l[j] == -1) ulist[pos++] = j;
        if (it >= 1) for (int j = g * EMD_THREADS + tid; j < n; j += G * EMD_THREADS) emd_st(top_z + j, 0ull);
        __syncthreads();

        // ---- Bid (:95-179): wave w of workgroup g takes bidders g*16+w, +16G, ...
        for (int u = g * EMD_WAVES + wave; u < U; u += G * EMD_WAVES) {
            const int i = ulist[u];
            const float x1 = p1[i * 3], y1 = p1[i * 3 + 1], z1 = p1[i * 3 + 2];
            Bid3 r{-1e9f, -1e9f, -1};                           // :116
            for (int k = lane; k < npad; k += 64 * EMD_UNROLL) {
#pragma unroll
                for (int e = 0; e < EMD_UNROLL; ++e) {
                    const int kk = k + 64 * e;
                    const float dx = tx[kk] - x1, dy = ty[kk] - y1, dz = tz[kk] - z1;              // :139-141
                    const float d = (3.0f - emd_sqrt(((dx * dx) + (dy * dy)) + (dz * dz))) - tp[kk];   // :143
                    r.idx = d > r.best ? kk : r.idx;                                               // :144-151
                    r.better = __builtin_amdgcn_fmed3f(r.best, d, r.better);
                    r.best = fmaxf(r.best, d);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(r.best, o, 64), obt = __shfl_xor(r.better, o, 64);
                const int oi = __shfl_xor(r.idx, o, 64);
                emd_merge(r, ob, obt, oi);
            }
            if (lane == 0) {
                const float v = (r.best - r.better) + eps;                                  // :175-176
                emd_st(bid_w + i, r.idx); emd_st(inc_w + i, v);
                const unsigned long long key =
                    ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)(0x7fffffff - i);
                __hip_atomic_fetch_max(top_w + r.idx, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // :177 + GetMax
            }
        }
        if (!(ok = emd_group_sync(counter, passed, G, &gave_up))) break;      // the round's only group barrier

        // ---- Assign (:196-215) of ALL bidders on this workgroup's copy: one lane per bidder
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int i = ulist[u];
            const int t = min(max(emd_ld(bid_w + i), 0), n - 1);
            if (last) { assign_l[i] = t; continue; }
            const unsigned long long key = emd_ld(top_w + t);
            const float v = emd_ld(inc_w + i);
            if (0x7fffffff - (int)(unsigned)key != i) continue;
            const int prev = inv_l[t];
            if (prev != -1) assign_l[prev] = -1;
            inv_l[t] = i;
            assign_l[i] = t;
            tp[t] = tp[t] + v;                                  // :211
        }
        __syncthreads();
    }

    if (!ok) {                                                  // the group barrier timed out: no result for this sample
        for (int j = g * EMD_THREADS + tid; j < n; j += G * EMD_THREADS) { dist[(size_t)b * n + j] = __builtin_nanf(""); assignment[(size_t)b * n + j] = -1; }
        return;
    }
    for (int j = g * EMD_THREADS + tid; j < n; j += G * EMD_THREADS) {        // CalcDist :217-226 + the assignment itself
        const int t = assign_l[j];
        assignment[(size_t)b * n + j] = t;
        if (t < 0 || t >= n) { dist[(size_t)b * n + j] = __builtin_nanf(""); continue; }
        const float dx = p1[j * 3] - tx[t], dy = p1[j * 3 + 1] - ty[t], dz = p1[j * 3 + 2] - tz[t];
        dist[(size_t)b * n + j] = ((dx * dx) + (dy * dy)) + (dz * dz);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// n <= EMD_GRID_MAX (the training call: n = SAMPLE_NUM * VP_NUM = 2048, train.py:193): the replicated-state rounds of
// emd_auction_local_kernel with a PRUNED Bid scan.  At B = 64 the auction is bound by the arithmetic of the scan
// (10 011 bidder scans x 2048 targets x 23 instructions per sample), and almost all of it is spent on targets that
// cannot matter: a bidder's value for target j is v_j = (3 - |x_i - y_j|) - price_j with price_j >= 0, so a target
// farther than R = 3 - S from the bidder (S = any lower bound of its final second-best value) is STRICTLY below the
// runner-up: it changes neither best, nor second best, nor a tie.  The targets do not move during an auction, so
// they are sorted once into a uniform 8 x 8 x 8 grid over their bounding box (counting sort in LDS; x-fastest cell
// order makes every (y, z) row of cells one contiguous run of sorted targets).  A bidder's wave
//   A. scans the 3 x 3 x 3 cells around the bidder (flattened into a per-wave list of target positions, so all 64 lanes
//      work even when a row holds a dozen targets) and merges: (best, second best, index);
//   B. takes S = that second best, R = 3 - S (+ margin), and if the cells covering [x - R, x + R]^3 go beyond block A,
//      scans that larger box instead (its result replaces A's).  Fewer than two targets in A: the whole grid.
// The cell of a coordinate is a monotone function of it and the box bounds go through the same function, so a target
// outside the box differs from the bidder by more than R along some axis: exact, whatever the rounding.
// Ties are resolved on ORIGINAL indices (the order inside a cell is whatever the counting sort's atomics gave, and
// differs between the workgroups of a sample; nothing depends on it).  Bit-equal to the oracle like the other two.
constexpr int EMD_GRID_MAX = 2048;
constexpr int EG = 8, ENC = EG * EG * EG;
constexpr int ELIST = 448;              // target positions of a box one wave flattens at a time

struct EmdGrid { float mn[3], sc[3]; };
__device__ inline int emd_cell1(float x, float mn, float sc) {
    const int c = (int)((x - mn) * sc);                          // monotone in x (truncation toward zero included)
    return min(max(c, 0), EG - 1);
}

// one target (sorted position kk) into the lane's running result; r.idx is a SORTED position here
__device__ inline void emd_eval(Bid3& r, int kk, float x1, float y1, float z1, const float* tx, const float* ty, const float* tz,
                                const float* tp, const int* orig) {
    const float dx = tx[kk] - x1, dy = ty[kk] - y1, dz = tz[kk] - z1;                                  // :139-141
    const float d = (3.0f - emd_sqrt(((dx * dx) + (dy * dy)) + (dz * dz))) - tp[kk];                    // :143
    bool take = d > r.best;                                                                         // :144-151
    if (__builtin_amdgcn_ballot_w64(d == r.best && r.idx >= 0))        // equal values: the lower ORIGINAL index wins (rare)
        take = take || (d == r.best && r.idx >= 0 && orig[kk] < orig[r.idx]);
    r.idx = take ? kk : r.idx;
    r.better = __builtin_amdgcn_fmed3f(r.best, d, r.better);
    r.best = fmaxf(r.best, d);
}

// wave64 reductions on the VALU (DPP row shifts + row broadcasts, result wave-uniform): a butterfly of ds_bpermute
// shuffles costs ~70 cycles per step and value, and a bidder's scan is a chain of latencies
#define EMD_DPPI(v, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, rmask, 0xf, false)
__device__ inline float emd_wave_max(float v) {
#define EMD_STEP(ctrl, rmask) v = fmaxf(v, __int_as_float(EMD_DPPI(__float_as_int(v), ctrl, rmask)))
    EMD_STEP(0x111, 0xf); EMD_STEP(0x112, 0xf); EMD_STEP(0x114, 0xf); EMD_STEP(0x118, 0xf); EMD_STEP(0x142, 0xa); EMD_STEP(0x143, 0xc);
#undef EMD_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline int emd_wave_min_i(int v) {
#define EMD_STEP(ctrl, rmask) v = min(v, EMD_DPPI(v, ctrl, rmask))
    EMD_STEP(0x111, 0xf); EMD_STEP(0x112, 0xf); EMD_STEP(0x114, 0xf); EMD_STEP(0x118, 0xf); EMD_STEP(0x142, 0xa); EMD_STEP(0x143, 0xc);
#undef EMD_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
// the 64 lanes' results over disjoint target sets (ORIGINAL indices) -> the result over their union, in every lane:
// largest value, lowest index among the lanes that hold it, second largest counting duplicates (= what folding the
// lanes with emd_merge gives: the winner's own runner-up competes with every other lane's best)
__device__ inline Bid3 emd_wave_merge(const Bid3& r) {
    Bid3 o;
    o.best = emd_wave_max(r.best);
    o.idx = emd_wave_min_i((r.best == o.best && r.idx >= 0) ? r.idx : 0x7fffffff);
    o.better = emd_wave_max((r.idx == o.idx && r.idx >= 0) ? r.better : r.best);
    if (o.idx == 0x7fffffff) o.idx = -1;
    return o;
}

// all targets of the cells [c0, c1] (per axis, inclusive) -> the merged result of the wave, with an ORIGINAL index
__device__ inline Bid3 emd_scan_box(const int c0[3], const int c1[3], float x1, float y1, float z1, const float* tx, const float* ty,
                                    const float* tz, const float* tp, const int* orig, const int* cell_start, int* wlist) {
    const int lane = threadIdx.x & 63;
    const int ny = c1[1] - c0[1] + 1, nz = c1[2] - c0[2] + 1, nrows = ny * nz;      // <= 64
    int s0 = 0, len = 0;
    if (lane < nrows) {
        const int cy = c0[1] + lane % ny, cz = c0[2] + lane / ny, base = (cz * EG + cy) * EG;
        s0 = cell_start[base + c0[0]];
        len = cell_start[base + c1[0] + 1] - s0;
    }
    int off = len;                                               // inclusive prefix sum over the rows
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(off, o, 64); if (lane >= o) off += t; }
    const int T = __shfl(off, 63, 64);
    Bid3 r{-1e9f, -1e9f, -1};                                    // :116
    // the rows' targets as ONE flat list, ELIST positions at a time: a lane writes the part of its row that falls into
    // the window, then all 64 lanes evaluate the window (a row holds a dozen targets: row by row, 50 lanes would idle)
    const int first = off - len;                                 // this lane's row covers flat positions [first, off)
    for (int w0 = 0; w0 < T; w0 += ELIST) {
        const int qlo = max(first, w0), qhi = min(off, w0 + ELIST);
        for (int q = qlo; q < qhi; ++q) wlist[q - w0] = s0 + (q - first);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int cw = min(ELIST, T - w0);
        for (int k = lane; k < cw; k += 64) emd_eval(r, wlist[k], x1, y1, z1, tx, ty, tz, tp, orig);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();                         // the list is rewritten by the next window / box
    }
    r.idx = r.idx >= 0 ? orig[r.idx] : -1;
    return emd_wave_merge(r);
}

__global__ __launch_bounds__(EMD_THREADS) void emd_auction_grid_kernel(const float* __restrict__ xyz1,
                                                                       const float* __restrict__ xyz2, int B, int n,
                                                                       int npad, int G, float eps, int iters,
                                                                       float* __restrict__ dist, int32_t* assignment,
                                                                       float* wsf, unsigned* counters) {
    extern __shared__ __attribute__((aligned(16))) float emd_lds[];      // 7 planes of npad floats + 5 arrays of npad ints
    float* tx = emd_lds; float* ty = tx + npad; float* tz = ty + npad; float* tp = tz + npad;       // SORTED by cell
    int* assign_l = reinterpret_cast<int*>(tp + npad);
    int* inv_l = assign_l + npad;
    int* ulist = inv_l + npad;
    int* orig = ulist + npad;           // sorted position -> target
    int* pos_of = orig + npad;          // target -> sorted position
    float* bx = reinterpret_cast<float*>(pos_of + npad);         // the bidders' coordinates: a global load per bidder was
    float* by = bx + npad; float* bz = by + npad;                // ~1 us at the head of every scan's latency chain
    __shared__ int cell_start[ENC + 8];
    __shared__ int cursor[ENC];
    __shared__ int wlists[EMD_WAVES][ELIST];
    __shared__ float red[EMD_WAVES][6];
    __shared__ EmdGrid grid;
    __shared__ int wcount[EMD_WAVES];
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / G) * 8 + xcd, g = q % G;
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * n * 3;
    float* base = wsf + (size_t)b * EMD_WS_PLANES * n;
    unsigned long long* top = reinterpret_cast<unsigned long long*>(base);
    int* gbid = reinterpret_cast<int*>(base + 6 * n);
    float* ginc = base + 8 * n;
    unsigned* counter = counters + 2 * b;
    unsigned passed = 0;
    bool ok = true;

    // ---- the grid: bounding box of the targets, counting sort by cell (once per auction)
    {
        float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (int j = tid; j < n; j += EMD_THREADS)
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float c = p2[(size_t)j * 3 + a]; lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c); }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
            if (lane == 0) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
        }
        for (int c = tid; c < ENC; c += EMD_THREADS) cursor[c] = 0;
        __syncthreads();
        if (tid < 3) {
            float l = red[0][tid], h = red[0][3 + tid];
            for (int w = 1; w < EMD_WAVES; ++w) { l = fminf(l, red[w][tid]); h = fmaxf(h, red[w][3 + tid]); }
            grid.mn[tid] = l;
            grid.sc[tid] = h > l ? (float)EG / (h - l) : 0.0f;      // a flat (or non-finite) extent: one layer of cells
        }
        __syncthreads();
        for (int j = tid; j < n; j += EMD_THREADS) {
            const int c = (emd_cell1(p2[(size_t)j * 3 + 2], grid.mn[2], grid.sc[2]) * EG + emd_cell1(p2[(size_t)j * 3 + 1], grid.mn[1], grid.sc[1])) * EG
                          + emd_cell1(p2[(size_t)j * 3], grid.mn[0], grid.sc[0]);
            atomicAdd(&cursor[c], 1);
        }
        __syncthreads();
        // exclusive scan of the ENC counts: lanes of the first ENC / 64 waves, then across waves
        int cnt = 0, incl = 0;
        if (tid < ENC) { cnt = cursor[tid]; incl = cnt; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wcount[wave] = incl;
        __syncthreads();
        if (tid < ENC) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += wcount[w];
            cell_start[tid] = before + incl - cnt;
            cursor[tid] = before + incl - cnt;
        }
        if (tid == 0) cell_start[ENC] = n;
        __syncthreads();
        for (int j = tid; j < n; j += EMD_THREADS) {
            const float x = p2[(size_t)j * 3], y = p2[(size_t)j * 3 + 1], z = p2[(size_t)j * 3 + 2];
            const int c = (emd_cell1(z, grid.mn[2], grid.sc[2]) * EG + emd_cell1(y, grid.mn[1], grid.sc[1])) * EG + emd_cell1(x, grid.mn[0], grid.sc[0]);
            const int pos = atomicAdd(&cursor[c], 1);
            tx[pos] = x; ty[pos] = y; tz[pos] = z; tp[pos] = 0.0f; orig[pos] = j; pos_of[j] = pos;
        }
        for (int j = tid; j < npad; j += EMD_THREADS) {
            if (j >= n) { tx[j] = 0.0f; ty[j] = 0.0f; tz[j] = 0.0f; tp[j] = __builtin_inff(); orig[j] = 0x7fffffff; }
            assign_l[j] = -1; inv_l[j] = -1;
            const int jc = min(j, n - 1);
            bx[j] = p1[(size_t)jc * 3]; by[j] = p1[(size_t)jc * 3 + 1]; bz[j] = p1[(size_t)jc * 3 + 2];
        }
    }
    for (int j = g * EMD_THREADS + tid; j < 3 * n; j += G * EMD_THREADS) emd_st(top + j, 0ull);
    ok = emd_group_sync(counter, passed, G, &gave_up);

    const int per = (n + EMD_THREADS - 1) / EMD_THREADS, j0 = min(n, tid * per), j1 = min(n, j0 + per);
    int* wlist = wlists[wave];
    for (int it = 0; ok && it < iters; ++it) {
        const bool last = it == iters - 1;
        unsigned long long* top_w = top + (size_t)(it % 3) * n;
        unsigned long long* top_z = top + (size_t)((it + 1) % 3) * n;
        int* bid_w = gbid + (size_t)(it & 1) * n;
        float* inc_w = ginc + (size_t)(it & 1) * n;
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += assign_l[j] == -1;
        int scan = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(scan, o, 64); if (lane >= o) scan += t2; }
        if (lane == 63) wcount[wave] = scan;
        __syncthreads();
        int before = 0, U = 0;
#pragma unroll
        for (int w = 0; w < EMD_WAVES; ++w) { const int c = wcount[w]; before += w < wave ? c : 0; U += c; }
        if (U == 0) break;
        int pos = before + scan - cnt;
        for (int j = j0; j < j1; ++j) if (assign_l[j] == -1) ulist[pos++] = j;
        if (it >= 1) for (int j = g * EMD_THREADS + tid; j < n; j += G * EMD_THREADS) emd_st(top_z + j, 0ull);
        __syncthreads();

        // ---- Bid (:95-179), pruned
        for (int u = g * EMD_WAVES + wave; u < U; u += G * EMD_WAVES) {
            const int i = ulist[u];
            const float x1 = bx[i], y1 = by[i], z1 = bz[i];
            const float xyz[3] = {x1, y1, z1};
            int a0[3], a1[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int c = emd_cell1(xyz[a], grid.mn[a], grid.sc[a]);
                a0[a] = max(c - 1, 0); a1[a] = min(c + 1, EG - 1);
            }
            Bid3 r = emd_scan_box(a0, a1, x1, y1, z1, tx, ty, tz, tp, orig, cell_start, wlist);
            // every target outside [x - R, x + R]^3 is strictly below the runner-up found so far
            const bool two = r.idx >= 0 && r.better > -1e8f;
            const float R = two ? (3.0f - r.better) + 1.0e-5f : __builtin_inff();
            int b0[3], b1[3];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                b0[a] = emd_cell1(xyz[a] - R, grid.mn[a], grid.sc[a]);
                b1[a] = emd_cell1(xyz[a] + R, grid.mn[a], grid.sc[a]);
                if (!(R < 1e30f)) { b0[a] = 0; b1[a] = EG - 1; }                // also a NaN radius: everything
                inside = inside && b0[a] >= a0[a] && b1[a] <= a1[a];
            }
            if (!inside) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { b0[a] = min(b0[a], a0[a]); b1[a] = max(b1[a], a1[a]); }
                r = emd_scan_box(b0, b1, x1, y1, z1, tx, ty, tz, tp, orig, cell_start, wlist);
            }
            if (lane == 0) {
                const float v = (r.best - r.better) + eps;                                  // :175-176
                emd_st(bid_w + i, r.idx); emd_st(inc_w + i, v);
                const unsigned long long key =
                    ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)(0x7fffffff - i);
                __hip_atomic_fetch_max(top_w + min(max(r.idx, 0), n - 1), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!(ok = emd_group_sync(counter, passed, G, &gave_up))) break;

        // ---- Assign (:196-215) of ALL bidders on this workgroup's copy
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int i = ulist[u];
            const int t = min(max(emd_ld(bid_w + i), 0), n - 1);
            if (last) { assign_l[i] = t; continue; }
            const unsigned long long key = emd_ld(top_w + t);
            const float v = emd_ld(inc_w + i);
            if (0x7fffffff - (int)(unsigned)key != i) continue;
            const int prev = inv_l[t];
            if (prev != -1) assign_l[prev] = -1;
            inv_l[t] = i;
            assign_l[i] = t;
            const int ps = pos_of[t];
            tp[ps] = tp[ps] + v;                                // :211
        }
        __syncthreads();
    }

    if (!ok) {
        for (int j = g * EMD_THREADS + tid; j < n; j += G * EMD_THREADS) { dist[(size_t)b * n + j] = __builtin_nanf(""); assignment[(size_t)b * n + j] = -1; }
        return;
    }
    for (int j = g * EMD_THREADS + tid; j < n; j += G * EMD_THREADS) {        // CalcDist :217-226 + the assignment itself
        const int t = assign_l[j];
        assignment[(size_t)b * n + j] = t;
        if (t < 0 || t >= n) { dist[(size_t)b * n + j] = __builtin_nanf(""); continue; }
        const int ps = pos_of[t];
        const float dx = bx[j] - tx[ps], dy = by[j] - ty[ps], dz = bz[j] - tz[ps];
        dist[(size_t)b * n + j] = ((dx * dx) + (dy * dy)) + (dz * dz);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the pruned auction again, rebuilt around two measurements.  (i) A late round of an easy auction (uniform
// clouds) is ~100 bidders per sample: a chain of latencies -- in emd_auction_grid_kernel ~19 us per round, ~6 of them five
// dependent L2 round trips of the atomic-max / counter / read-back exchange.  (ii) The auction the training step really
// runs (points on K small primitives against a cloud that fills the cube, train.py:193 early in training) keeps 500-1500
// bidders per round and their radius grows to 2-3 cells: there that kernel is bound by VALU issue -- ~1000 wave-instructions
// per bidder, of which ~80 targets (the ones inside the radius) out of ~300 scanned (the cell box) matter.
//   * TARGET IDS ARE SORTED POSITIONS.  The counting sort is deterministic (rank inside a cell = rank of the original
//     index): every workgroup of a sample holds the same order; owner table, per-target maxima, prices and bids speak
//     sorted positions, the original index only breaks ties and is restored in the output.
//   * STATIC OWNERSHIP: point i always bids from workgroup i mod G, so its coordinates and its MEMORY -- the two targets
//     that were best and second best in its last bid -- live in that workgroup's LDS.  Any two distinct targets bound the
//     second-best value from below (min of their two current values), hence the radius R = 3 - that bound holds every
//     target that can matter NOW: one scan per bid, with a radius that is exact unless a remembered target was repriced.
//   * TEAMS: T lanes per bidder, T = the power of two that spreads the round's own bidders over ~1024 lanes (at most
//     16).  A team walks the (y, z) rows of a 32 x 8 x 8 grid inside the disc of radius R around the bidder and, per row,
//     the cells of the chord: the scanned set is the sphere at cell resolution, not its bounding box of cells.  Rounds with
//     much work bid in the BALANCED FORM instead (emd_flat_bid below: the rows of all own bidders in one list sorted by
//     length, dealt over the lanes; price filter; best two by LDS atomic maxima).
//   * EXCHANGE BY TAGGED GRANULES (the hand-off form MI355X_MICROARCH.md prices at ~1 us: one naturally aligned 8-byte
//     {tag | target | increment} written by ONE sc1 store, polled with sc1 loads): every workgroup knows the round's
//     bidder list, so it knows which entries to wait for; no counter, no barrier, no atomics in memory.  GetMax
//     (emd_cuda.cu:181-194) is an LDS 64-bit atomic max per target on every workgroup's own copy.
//   * 16-bit state: 32 n + 18 n / G bytes of LDS + the cell table + the balanced form's lists (51.7 KB): ONE workgroup
//     per CU with up to 128 VGPRs (both forms of the Bid phase inline without scratch), G = 4 at B = 64; the 32 KB of LDS
//     and the wave slots it leaves are what the training step's other kernels run in beside it.
// Same arithmetic, same tie rules: bit-equal to the oracle and to the other three kernels for every group size.
#ifndef EMD_EG
#define EMD_EG 8
#endif
#ifndef EMD_EGX
#define EMD_EGX 32
#endif
constexpr int EG3 = EMD_EG, EGX3 = EMD_EGX, ENC3 = EG3 * EG3 * EGX3;     // cells along y and z; at most EGX3 along x (the contiguous axis)
constexpr unsigned EMD_MEM_NONE = 0xffffffffu;

struct EmdGrid3 { float mn[3], sc[3], cw[3], mg; int egx; };     // origin, cells per unit, cell width; geometric margin; cells along x
__device__ inline int emd_cell3(float x, float mn, float sc, int cells = EG3) {
    const int c = (int)((x - mn) * sc);                          // monotone in x (truncation toward zero included)
    return min(max(c, 0), cells - 1);
}

__device__ inline int emd_wave_scan_incl(int v) {          // inclusive prefix sum over the 64 lanes, on the VALU
#define EMD_SSTEP(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)
    EMD_SSTEP(0x111, 0xf); EMD_SSTEP(0x112, 0xf); EMD_SSTEP(0x114, 0xf); EMD_SSTEP(0x118, 0xf);
    EMD_SSTEP(0x142, 0xa); EMD_SSTEP(0x143, 0xc);
#undef EMD_SSTEP
    return v;
}

// a lane's (team's) running result: the two largest values (counting duplicates) and the sorted positions that hold them
struct Top2 { float b, s; int i, j; };

__device__ inline float emd_value(const float4 t, float x1, float y1, float z1) {
    const float dx = t.x - x1, dy = t.y - y1, dz = t.z - z1;                                          // :139-141
    return (3.0f - emd_sqrt(((dx * dx) + (dy * dy)) + (dz * dz))) - t.w;                               // :143
}
// c ? a : b as ONE v_cndmask on the compare's lane mask: left to itself the compiler turns the three dependent selects of
// top2_put into nested exec-mask branches (a dozen scalar instructions per target in a loop bound by instruction issue)
__device__ inline int emd_sel(bool c, int a, int b) {
    int r;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
    __asm__("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ inline void top2_put(Top2& r, int k, float d) {                                            // :144-151
    const bool gb = d > r.b, gs = d > r.s;
    r.j = emd_sel(gb, r.i, emd_sel(gs, k, r.j));
    r.s = __builtin_amdgcn_fmed3f(r.b, d, r.s);
    r.i = emd_sel(gb, k, r.i);
    r.b = fmaxf(r.b, d);
}
// fold the result over a DISJOINT target set (o) into a
__device__ inline void top2_merge(Top2& a, const Top2& o) {
    const bool gb = a.b > o.b;
    const float lb = fminf(a.b, o.b), ws = gb ? a.s : o.s;
    const int li = gb ? o.i : a.i, wj = gb ? a.j : o.j;
    a.i = gb ? a.i : o.i;
    a.b = fmaxf(a.b, o.b);
    a.j = ws >= lb ? wj : li;
    a.s = fmaxf(ws, lb);
}
template <int CTRL>
__device__ inline void top2_step_dpp(Top2& r) {
    Top2 o;
    o.b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r.b), CTRL, 0xf, 0xf, false));
    o.s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r.s), CTRL, 0xf, 0xf, false));
    o.i = __builtin_amdgcn_update_dpp(0, r.i, CTRL, 0xf, 0xf, false);
    o.j = __builtin_amdgcn_update_dpp(0, r.j, CTRL, 0xf, 0xf, false);
    top2_merge(r, o);
}
__device__ inline void top2_step_xor(Top2& r, int m) {
    Top2 o;
    o.b = __shfl_xor(r.b, m, 64); o.s = __shfl_xor(r.s, m, 64); o.i = __shfl_xor(r.i, m, 64); o.j = __shfl_xor(r.j, m, 64);
    top2_merge(r, o);
}
// the T lanes of a team (aligned, T a power of two) -> the team's result in every one of them.  Every step folds two
// disjoint halves whose lanes all hold their half's result: quad permutes, then the mirrors of 8 and 16 lanes, then
// the cross-row exchanges.
__device__ inline void top2_team(Top2& r, int T) {
    if (T >= 2) top2_step_dpp<0xB1>(r);        // quad_perm [1,0,3,2]
    if (T >= 4) top2_step_dpp<0x4E>(r);        // quad_perm [2,3,0,1]
    if (T >= 8) top2_step_dpp<0x141>(r);       // row_half_mirror
    if (T >= 16) top2_step_dpp<0x140>(r);      // row_mirror
    if (T >= 32) top2_step_xor(r, 16);
    if (T >= 64) top2_step_xor(r, 32);
}
__device__ inline int emd_team_min(int v, int T) {
#define EMD_MSTEP(ctrl) v = min(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
    if (T >= 2) EMD_MSTEP(0xB1);
    if (T >= 4) EMD_MSTEP(0x4E);
    if (T >= 8) EMD_MSTEP(0x141);
    if (T >= 16) EMD_MSTEP(0x140);
#undef EMD_MSTEP
    if (T >= 32) v = min(v, __shfl_xor(v, 16, 64));
    if (T >= 64) v = min(v, __shfl_xor(v, 32, 64));
    return v;
}

// a value as an unsigned key of the same order (and back); 0 is no value
__device__ inline unsigned emd_ord(float v) { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float emd_unord(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
// the price filter of the Bid scans: can target t reach a value of 3 - R for the bidder at (x1, y1, z1)?  d <= R - price,
// tested on the squares (no square root; R carries the auction's `slack`, many times the rounding of this test)
__device__ inline bool emd_may_matter(const float4 t, float x1, float y1, float z1, float R) {
    const float dx = t.x - x1, dy = t.y - y1, dz = t.z - z1;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float q = R - t.w;
    return d2 <= q * fabsf(q);                                    // q < 0 (the price alone is too high): -q^2 < d2
}

struct EmdTables3 {
    const float4* t4;                 // targets (x, y, z, price), sorted by cell then index
    const unsigned short* orig;
    const unsigned short* cell_start;
    const EmdGrid3* grid;
#ifdef EMD_TRACE
    unsigned *tr_evals, *tr_rows;
#endif
};

// Every target within R of (x1, y1, z1) -- and whatever else shares its cells -- is handed to f(k) exactly once over
// the T lanes of the team (tl = this lane's number in it): the (y, z) rows of cells inside the disc of radius R around
// the bidder are dealt to the lanes (row r of the disc's bounding square to lane r mod T; L lanes share a row when the
// team has more lanes than the square has rows), per row the cells of the chord -- the cells along x are fine (up to 32)
// because trimming a contiguous run costs nothing.  The row bounds come from the same monotone cell function the sort
// used, a row's distance from the cell boundaries minus a margin that covers their rounding: a target left out is farther
// than R.  (Measured against the alternative -- all lanes of a team walk every row together, striding over its targets --
// on 16^3 and 32 x 8 x 8 grids: this form on 32 x 8 x 8 is the fastest on crowded and on sparse auctions.)
template <typename F>
__device__ inline void emd_walk(float x1, float y1, float z1, float R, int tl, int T, const EmdTables3& Tb, F&& f) {
    const EmdGrid3& g = *Tb.grid;
    const int cy0 = emd_cell3(y1 - R, g.mn[1], g.sc[1]), cy1 = emd_cell3(y1 + R, g.mn[1], g.sc[1]);
    const int cz0 = emd_cell3(z1 - R, g.mn[2], g.sc[2]), cz1 = emd_cell3(z1 + R, g.mn[2], g.sc[2]);
    const int ny = cy1 - cy0 + 1, nrows = ny * (cz1 - cz0 + 1);
    const float inv_ny = 1.0f / (float)ny;
    int L = 1, lg = 0;
    while (2 * L * nrows <= T) { L *= 2; ++lg; }
    const int RS = T >> lg, sub = tl & (L - 1);
    const float R2 = R * R;
    for (int r = tl >> lg; r < nrows; r += RS) {
        const int rz = (int)(((float)r + 0.5f) * inv_ny), ry = r - rz * ny;
        const int cy = cy0 + ry, cz = cz0 + rz;
        // distance of the row from the bidder in y and z: below its lower edge or above its upper one (cells 0 and EG3 - 1
        // also hold what the clamp folded into them: their outer edges are the cloud's own bounds), else inside: 0
        const float ylo = g.mn[1] + (float)cy * g.cw[1], zlo = g.mn[2] + (float)cz * g.cw[2];
        const float dy = fmaxf(fmaxf(ylo - y1, y1 - (ylo + g.cw[1])) - g.mg, 0.0f);
        const float dz = fmaxf(fmaxf(zlo - z1, z1 - (zlo + g.cw[2])) - g.mg, 0.0f);
        const float h2 = (R2 - dy * dy) - dz * dz;
        if (!(h2 >= 0.0f)) continue;
        const float h = __builtin_amdgcn_sqrtf(h2) * 1.000001f + g.mg;
        const int cx0 = emd_cell3(x1 - h, g.mn[0], g.sc[0], g.egx), cx1 = emd_cell3(x1 + h, g.mn[0], g.sc[0], g.egx);
        const int base = (cz * EG3 + cy) * g.egx;
        int k = (int)Tb.cell_start[base + cx0] + sub;
        const int e = (int)Tb.cell_start[base + cx1 + 1];
#ifdef EMD_TRACE
        if (Tb.tr_rows) { atomicAdd(Tb.tr_rows, 1u); if (e > k) atomicAdd(Tb.tr_evals, (unsigned)((e - k + L - 1) / L)); }
#endif
        for (; k < e; k += L) f(k);
    }
}

// one (y, z) row of cells against the ball of radius R (R2 = R * R) around the bidder: the sorted positions [kst, kst + cnt)
// of the row's chord -- the arithmetic of emd_walk, for the balanced form of the Bid phase
__device__ inline void emd_row_chord(float x1, float y1, float z1, float R2, int cy, int cz, const EmdGrid3& g,
                                     const unsigned short* cell_start, int& kst, int& cnt) {
    const float ylo = g.mn[1] + (float)cy * g.cw[1], zlo = g.mn[2] + (float)cz * g.cw[2];
    const float dy = fmaxf(fmaxf(ylo - y1, y1 - (ylo + g.cw[1])) - g.mg, 0.0f);
    const float dz = fmaxf(fmaxf(zlo - z1, z1 - (zlo + g.cw[2])) - g.mg, 0.0f);
    const float h2 = (R2 - dy * dy) - dz * dz;
    kst = 0; cnt = 0;
    if (!(h2 >= 0.0f)) return;
    const float h = __builtin_amdgcn_sqrtf(h2) * 1.000001f + g.mg;
    const int cx0 = emd_cell3(x1 - h, g.mn[0], g.sc[0], g.egx), cx1 = emd_cell3(x1 + h, g.mn[0], g.sc[0], g.egx);
    const int base = (cz * EG3 + cy) * g.egx;
    kst = (int)cell_start[base + cx0];
    cnt = (int)cell_start[base + cx1 + 1] - kst;
}

// state of emd_auction_team_kernel that its out-of-line part shares (static LDS; the dynamic part is laid out from npad, lgG)
__shared__ unsigned short emd3_cell_start[ENC3 + 8];
__shared__ EmdGrid3 emd3_grid;
__shared__ unsigned emd3_wtot[16];
extern __shared__ __attribute__((aligned(16))) float emd3_dyn[];
#ifdef EMD_TRACE
__shared__ unsigned emd3_tr_evals, emd3_tr_rows;
#endif

// BALANCED FORM (rounds with many bidders).  In a team a lane's work is whole rows of its own bidder and a wave
// waits for its heaviest lane: ~2.2x the mean on the step's clouds (tools/emd_balance_sim.py).  Here the rows of
// ALL own bidders become one list -- (first position, count, bidder slot) per row --, the list is counting-sorted
// by count, and lane t takes the rows of rank t, 2047 - t, 2048 + t, ...: the 64 lanes of a wave hold rows of
// (nearly) the same length, every lane gets long and short ones.  The lanes no longer belong to a bidder, so the
// survivors of the price filter go to the bidder's best / second through two LDS 64-bit atomic maxima of
// (value, lowest original index first, position): `old = max(best, key); max(second, min(old, key))` leaves the
// largest key in best and the largest of the rest in second whatever the order -- the same two values the
// teams find, and the index rule of the tie pass for free.
// Its lists have LDS of their own behind the auction's state (one workgroup per CU: 160 KB).  A batch is up to
// EMD_FLAT_SLOTS bidders and EMD_FLAT_ROWS rows: one batch per round at G >= 4.
constexpr int EMD_FLAT_SLOTS = 512, EMD_FLAT_ROWS = 8192, EMD_FLAT_RPL = EMD_FLAT_ROWS / EMD_THREADS;
constexpr int EMD_FLAT_BYTES = 8 + 32 * EMD_FLAT_SLOTS + 4 * EMD_FLAT_ROWS + 2 * EMD_FLAT_SLOTS + 2 * (EMD_FLAT_SLOTS + 4) + 512;   // 8: alignment of the first list
static_assert(EMD_FLAT_BYTES % 8 == 0 && EMD_FLAT_SLOTS <= 512 && EMD_FLAT_ROWS <= 65535, "nine slot bits in a row entry; 16-bit row numbers");
__device__ inline unsigned short* emd3_flat_frow(int npad, int lgG);
__shared__ int emd3_flat_off;        // a bidder lost its memory (degenerate clouds): its ball has to grow again, the teams do that
__shared__ float emd3_flat_tpb;      // targets per bid in the rows of the last balanced round

// r_first > 0: the FIRST bids of the auction (nobody has a memory yet): every bidder scans the ball of that radius; one
// that finds two targets in it has found its best two (everything outside is farther than both, and no price is
// negative), the others -- returned as a list in `frow`, their number as the result -- are left to the teams, whose ball
// grows.  -1: more bidders or rows than one batch holds, nothing was done.
__device__ __forceinline__ int emd_flat_bid(int npad, int lgG, int Uown, int n, float eps, float slack, unsigned tag,
                                            unsigned long long* bid_w, unsigned* ftrace, float r_first) {
    const int nown = npad >> lgG;
    float4* t4 = reinterpret_cast<float4*>(emd3_dyn);
    unsigned long long* top_l = reinterpret_cast<unsigned long long*>(t4 + npad);
    float* ox = reinterpret_cast<float*>(top_l + npad); float* oy = ox + nown; float* oz = oy + nown;
    unsigned* mem = reinterpret_cast<unsigned*>(oz + nown);
    unsigned short* orig = reinterpret_cast<unsigned short*>(mem + nown);
    unsigned short* ulist = orig + 3 * (size_t)npad;             // behind assign_l and inv_l
    unsigned short* ownu = ulist + npad;
    const unsigned short* cell_start = emd3_cell_start;
    const EmdGrid3& grid = emd3_grid;
    unsigned* const wtot = emd3_wtot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(EG3 <= 8, "three bits per cell coordinate in fbox");
    constexpr int NB = EMD_FLAT_SLOTS;
    // (18 nown bytes of own-bidder state end on an 8-byte boundary only when nown is not a multiple of 8: float4 wants 16)
    float4* fxyzr = reinterpret_cast<float4*>(reinterpret_cast<char*>(ownu + nown) + ((nown & 7) ? 8 : 0));   // [NB] the slot's bidder and its radius (nown is a multiple of 4)
    unsigned long long* fbest = reinterpret_cast<unsigned long long*>(fxyzr + NB);// [NB]
    unsigned long long* fsecond = fbest + NB;                                     // [NB]
    unsigned* fsorted = reinterpret_cast<unsigned*>(fsecond + NB);                // [EMD_FLAT_ROWS] rows by count, descending
    unsigned short* fbox = reinterpret_cast<unsigned short*>(fsorted + EMD_FLAT_ROWS);   // [NB] cy0 | cz0 << 3 | (ny - 1) << 6
    unsigned short* frow = fbox + NB;                                             // [NB + 4] first row of the slot
    unsigned* fhist = reinterpret_cast<unsigned*>(frow + NB + 4);                 // [64] rows per count
    unsigned* fcur = fhist + 64;                                                  // [64]
    __shared__ int f_nb, f_nr, f_nz, f_left;
    __shared__ unsigned f_nc;
    const bool first = r_first > 0.0f;
    if (tid == 0) f_left = 0;                            // targets in the round's rows: the work the next round's choice of form goes by
    if (tid == 0) f_nc = 0u;
    for (int k0 = 0; k0 < Uown;) {
        // -- A: one lane per bidder: radius from its memory, the square of rows around it
        int nrows = 0, kb = k0 + tid;
        const bool slot = tid < NB && kb < Uown;
        if (tid == 0) { f_nb = 0; f_nr = 0; }
        if (tid < 64) fhist[tid] = 0u;
        if (tid < NB) { fbest[tid] = 0ull; fsecond[tid] = 0ull; }
        if (slot) {
            const int l = (int)ulist[ownu[kb]] >> lgG;
            const float x1 = ox[l], y1 = oy[l], z1 = oz[l];
            float R = r_first;
            if (!first) {
                const unsigned m = mem[l];
                const float v1 = emd_value(t4[m & 0xffffu], x1, y1, z1), v2 = emd_value(t4[m >> 16], x1, y1, z1);
                R = (3.0f - fminf(v1, v2)) + slack;
            }
            if (!(R < 1e30f)) R = __builtin_inff();
            const int cy0 = emd_cell3(y1 - R, grid.mn[1], grid.sc[1]), cy1 = emd_cell3(y1 + R, grid.mn[1], grid.sc[1]);
            const int cz0 = emd_cell3(z1 - R, grid.mn[2], grid.sc[2]), cz1 = emd_cell3(z1 + R, grid.mn[2], grid.sc[2]);
            const int ny = cy1 - cy0 + 1;
            nrows = ny * (cz1 - cz0 + 1);
            fxyzr[tid] = make_float4(x1, y1, z1, R);
            fbox[tid] = (unsigned short)(cy0 | (cz0 << 3) | ((ny - 1) << 6));
        }
        const int incl = emd_wave_scan_incl(nrows);
        if (lane == 63) wtot[wave] = (unsigned)incl;
        __syncthreads();
        if (tid < NB) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += (int)wtot[w];
            const int end = before + incl;
            frow[tid] = (unsigned short)min(end - nrows, 0xffff);      // lanes past the last bidder: the total (the end of the last slot)
            if (tid == NB - 1) frow[NB] = (unsigned short)min(end, 0xffff);
            const bool fits = slot && end <= EMD_FLAT_ROWS;      // a prefix of the slots: the batch
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(fits);
            if (bal && lane == 63 - __builtin_clzll(bal)) { atomicAdd(&f_nb, __builtin_popcountll(bal)); atomicMax(&f_nr, end); }
        }
        __syncthreads();
        const int nb = f_nb, NR = f_nr;
        if (nb <= 0) break;                              // cannot happen (a bidder has at most 64 rows); never spin on it: the partners' bounded wait reports the sample
        if (first && nb < Uown) return -1;               // the first round in one batch or not at all (nothing has been bid yet)
#ifdef EMD_TRACE
        if (ftrace) { ftrace[0] = (unsigned)__builtin_amdgcn_s_memrealtime(); ftrace[3] = (k0 == 0 ? 0u : ftrace[3]) + 0x1000000u; }
#endif
        // -- B: the rows' chords, EMD_THREADS lanes over NR rows in runs of `rpl` consecutive ones (one search for the
        //       slot per lane, then steps); histogram of the counts
        unsigned ent[EMD_FLAT_RPL];
        int tr_nc = 0;
        {
            const int rpl = (NR + EMD_THREADS - 1) / EMD_THREADS;
            int f = tid * rpl, lo = 0, ry = 0, rz = 0, ny = 1, cy0 = 0, cz0 = 0, nxt = 0;
            float4 bd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (f < NR) {
                int hi = nb;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)frow[mid] <= f) lo = mid; else hi = mid; }
                const int r = f - (int)frow[lo], box = fbox[lo];
                ny = ((box >> 6) & 7) + 1; cy0 = box & 7; cz0 = (box >> 3) & 7;
                rz = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)ny)); ry = r - rz * ny;   // r + 0.5 is >= 1/16 from a multiple of ny
                nxt = frow[lo + 1];
                bd = fxyzr[lo];
            }
#ifdef EMD_TRACE_B2
            if (ftrace) ftrace[0] = (unsigned)__builtin_amdgcn_s_memrealtime();     // experiment: A column = up to the end of wave 0's slot search
#endif
#pragma unroll
            for (int j = 0; j < EMD_FLAT_RPL; ++j, ++f) {
                ent[j] = 0u;
                if (j >= rpl || f >= NR) continue;
                if (f == nxt) {                                  // the next slot's first row (every slot has rows)
                    ++lo;
                    const int box = fbox[lo];
                    ny = ((box >> 6) & 7) + 1; cy0 = box & 7; cz0 = (box >> 3) & 7; ry = 0; rz = 0;
                    nxt = frow[lo + 1];
                    bd = fxyzr[lo];
                }
                int kst, cnt;
                emd_row_chord(bd.x, bd.y, bd.z, bd.w * bd.w, cy0 + ry, cz0 + rz, grid, cell_start, kst, cnt);
                if (++ry == ny) { ry = 0; ++rz; }
                if (cnt > 0) {
                    ent[j] = (unsigned)kst | ((unsigned)cnt << 11) | ((unsigned)lo << 23);
                    atomicAdd(&fhist[min(cnt, 63)], 1u);
                    tr_nc += cnt;
                }
            }
        }
#ifdef EMD_TRACE_B2
        if (ftrace) ftrace[1] = (unsigned)__builtin_amdgcn_s_memrealtime();         // experiment: B column = wave 0's rows (no barrier)
#endif
        { const int t = emd_wave_scan_incl(tr_nc); if (lane == 63 && t) atomicAdd(&f_nc, (unsigned)t); }
        __syncthreads();
#ifdef EMD_TRACE_B
        if (ftrace) ftrace[0] = (unsigned)__builtin_amdgcn_s_memrealtime();     // experiment: A column = up to the end of B's chords
#endif
        if (wave == 0) {                                 // longest rows first
            const int h = (int)fhist[63 - lane];
            const int in = emd_wave_scan_incl(h);
            fcur[63 - lane] = (unsigned)(in - h);
            if (lane == 63) f_nz = in;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EMD_FLAT_RPL; ++j)
            if (ent[j]) fsorted[atomicAdd(&fcur[min((int)((ent[j] >> 11) & 4095u), 63)], 1u)] = ent[j];
        __syncthreads();
        // -- C: the rows, dealt back and forth over the lanes
        const int NZ = f_nz;
#ifdef EMD_TRACE
#ifndef EMD_TRACE_B2
        if (ftrace) ftrace[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
        if (ftrace) ftrace[3] += (unsigned)NR;
#endif
        // one row (or a lane's share of one): positions k, k + ST, ... below kend of the bidder in slot sl.  Four targets per
        // trip, their loads in flight together, no branches: the results are lane masks, the tail of a row is a mask too.  A
        // load past the row's end reads the next cells or, past n, whatever follows the targets in LDS: it is masked -- a
        // target outside its row's chord must not be offered, its own row offers it.
        auto scan = [&](const int ST, int k, int kend, int sl, float x1, float y1, float z1, float R) __attribute__((always_inline)) {
            int pend = -1;
            unsigned long long has = 0ull;               // lanes with a parked survivor
            auto offer = [&](int kk) {
                const float v = emd_value(t4[kk], x1, y1, z1);
                const unsigned long long key = ((unsigned long long)emd_ord(v) << 32) |
                                               ((unsigned long long)(0xffffu - (unsigned)orig[kk]) << 16) | (unsigned)kk;
                const unsigned long long old = atomicMax(&fbest[sl], key);
                atomicMax(&fsecond[sl], old < key ? old : key);
            };
            auto park = [&](unsigned long long pm, int kk) {
                if (pm & has) {                          // some lane found its second one: the wave evaluates the parked ones
                    if (pend >= 0) { offer(pend); pend = -1; }
                    has = 0ull;
                }
                __asm__("v_cndmask_b32 %0, %1, %2, %3" : "=v"(pend) : "v"(pend), "v"(kk), "s"(pm));
                has |= pm;
            };
            for (; k < kend; k += 4 * ST) {
                const float4* c = t4 + k;
                const float4 c0 = c[0], c1 = c[ST], c2 = c[2 * ST], c3 = c[3 * ST];
                const int left = kend - k;
                const unsigned long long m0 = __builtin_amdgcn_ballot_w64(emd_may_matter(c0, x1, y1, z1, R));
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(emd_may_matter(c1, x1, y1, z1, R)) & __builtin_amdgcn_ballot_w64(left > ST);
                const unsigned long long m2 = __builtin_amdgcn_ballot_w64(emd_may_matter(c2, x1, y1, z1, R)) & __builtin_amdgcn_ballot_w64(left > 2 * ST);
                const unsigned long long m3 = __builtin_amdgcn_ballot_w64(emd_may_matter(c3, x1, y1, z1, R)) & __builtin_amdgcn_ballot_w64(left > 3 * ST);
                park(m0, k); park(m1, k + ST); park(m2, k + 2 * ST); park(m3, k + 3 * ST);
            }
            if (pend >= 0) offer(pend);
        };
        // rows of 63 targets and more (the last bin: the head of the list) are not a lane's job -- a cloud that sits on
        // surfaces puts hundreds of targets into one row of cells: a WAVE takes each, its lanes striding through it
        const int NL = (int)fhist[63];
        for (int r = wave; r < NL; r += EMD_WAVES) {
            const unsigned e = fsorted[r];
            const int kst = (int)(e & 2047u), sl = (int)(e >> 23);
            const float4 bd = fxyzr[sl];
            scan(64, kst + lane, kst + (int)((e >> 11) & 4095u), sl, bd.x, bd.y, bd.z, bd.w);
        }
        // the other rows: one lane each, dealt back and forth over the lanes -- or, when the round has fewer rows than lanes
        // (a late round, a cloud on surfaces: there the round is as long as its longest row), 2, 4, ... 64 lanes each
        int lgS = 0;
        while (lgS < 6 && ((NZ - NL) << (lgS + 1)) <= EMD_THREADS) ++lgS;
#ifdef EMD_TRACE
        const unsigned long long tc0 = __builtin_readcyclecounter();
#endif
        if (lgS == 0) {
            for (int p = 0; NL + p * EMD_THREADS < NZ; ++p) {
                const int rank = NL + p * EMD_THREADS + ((p & 1) ? EMD_THREADS - 1 - tid : tid);
                int k = 0, kend = 0, sl = 0;
                float x1 = 0.0f, y1 = 0.0f, z1 = 0.0f, R = -1.0f;
                if (rank < NZ) {
                    const unsigned e = fsorted[rank];
                    k = (int)(e & 2047u); kend = k + (int)((e >> 11) & 4095u); sl = (int)(e >> 23);
                    const float4 bd = fxyzr[sl];
                    x1 = bd.x; y1 = bd.y; z1 = bd.z; R = bd.w;
                }
                scan(1, k, kend, sl, x1, y1, z1, R);
            }
        } else {
            const int rank = NL + (tid >> lgS);
            int k = 0, kend = 0, sl = 0;
            float x1 = 0.0f, y1 = 0.0f, z1 = 0.0f, R = -1.0f;
            if (rank < NZ) {
                const unsigned e = fsorted[rank];
                k = (int)(e & 2047u); kend = k + (int)((e >> 11) & 4095u); sl = (int)(e >> 23);
                k += tid & ((1 << lgS) - 1);
                const float4 bd = fxyzr[sl];
                x1 = bd.x; y1 = bd.y; z1 = bd.z; R = bd.w;
            }
            scan(1 << lgS, k, kend, sl, x1, y1, z1, R);
        }
#ifdef EMD_TRACE
        if (tid == 0) emd3_tr_evals += (unsigned)(__builtin_readcyclecounter() - tc0);
#endif
        __syncthreads();
#ifdef EMD_TRACE
        if (ftrace) { ftrace[2] = (unsigned)__builtin_amdgcn_s_memrealtime(); ftrace[3] = (ftrace[3] & 0xff00ffffu) | (min(f_nc >> 10, 255u) << 16); }
#endif
        // -- D: the bids
        if (tid < nb) {
            const int u = ownu[k0 + tid], l = (int)ulist[u] >> lgG;
            const unsigned long long kb1 = fbest[tid], kb2 = fsecond[tid];
            Top2 r{-1e9f, -1e9f, -1, -1};
            if (kb1) { r.b = emd_unord((unsigned)(kb1 >> 32)); r.i = (int)(kb1 & 0xffffu); }
            if (kb2) { r.s = emd_unord((unsigned)(kb2 >> 32)); r.j = (int)(kb2 & 0xffffu); }
            const bool two = r.i >= 0 && r.j >= 0 && r.i != r.j;
            if (first && !two) {
                frow[atomicAdd(&f_left, 1)] = (unsigned short)(k0 + tid);               // (frow is done with: the rows are sorted)
            } else {
                mem[l] = two ? ((unsigned)r.i | ((unsigned)r.j << 16)) : EMD_MEM_NONE;
                if (!two) emd3_flat_off = 1;
                const float v = (r.b - r.s) + eps;                                      // :175-176
                const unsigned t = (unsigned)min(max(r.i, 0), n - 1);
                emd_st(bid_w + u, ((unsigned long long)((tag << 16) | t) << 32) | (unsigned)__float_as_int(v));
            }
        }
        k0 += nb;
        if (k0 < Uown) __syncthreads();                  // the next batch reuses the lists
    }
    if (first) {
        __syncthreads();                                 // the list of the bidders left over is complete
        return f_left;
    }
    if (tid == 0) emd3_flat_tpb = (float)f_nc / (float)Uown;     // (every batch's B phase, hence its f_nc, lies before a barrier)
    return 0;
}

// where emd_flat_bid keeps `frow` (its list of left-over bidders after a first round): the same carving as above
__device__ inline unsigned short* emd3_flat_frow(int npad, int lgG) {
    const int nown = npad >> lgG;
    char* base = reinterpret_cast<char*>(emd3_dyn) + (size_t)npad * 32 + (size_t)nown * 18 + ((nown & 7) ? 8 : 0);   // behind ownu, aligned
    return reinterpret_cast<unsigned short*>(base + 32 * EMD_FLAT_SLOTS + 4 * EMD_FLAT_ROWS + 2 * EMD_FLAT_SLOTS);
}

// The team form of the Bid phase
__device__ __forceinline__ void emd_team_bid(int npad, int lgG, int Uown, int n, float eps, float slack, float r0,
                                             unsigned tag, unsigned long long* bid_w, int tnum, int tmax,
                                             const unsigned short* sub = nullptr) {   // sub: Uown positions in ownu (the bidders a first balanced round left over)
    const int nown = npad >> lgG;
    float4* t4 = reinterpret_cast<float4*>(emd3_dyn);
    unsigned long long* top_l = reinterpret_cast<unsigned long long*>(t4 + npad);
    float* ox = reinterpret_cast<float*>(top_l + npad); float* oy = ox + nown; float* oz = oy + nown;
    unsigned* mem = reinterpret_cast<unsigned*>(oz + nown);
    unsigned short* orig = reinterpret_cast<unsigned short*>(mem + nown);
    unsigned short* ulist = orig + 3 * (size_t)npad;             // behind assign_l and inv_l
    unsigned short* ownu = ulist + npad;
    const int tid = threadIdx.x;
#ifdef EMD_TRACE
    const EmdTables3 Tb{t4, orig, emd3_cell_start, &emd3_grid, &emd3_tr_evals, &emd3_tr_rows};
#else
    const EmdTables3 Tb{t4, orig, emd3_cell_start, &emd3_grid};
#endif
    int T = 1, lgT = 0;
    while (T < tmax && 2 * T * Uown <= tnum) { T *= 2; ++lgT; }
    const int per_pass = EMD_THREADS >> lgT, tl = tid & (T - 1);
    for (int k0 = 0; k0 < Uown; k0 += per_pass) {
        const int kb = k0 + (tid >> lgT);
        const bool have = kb < Uown;
        int u = 0, l = 0;
        float x1 = 0.0f, y1 = 0.0f, z1 = 0.0f, R = -1.0f;
        unsigned m = EMD_MEM_NONE;
        if (have) {
            u = ownu[sub ? (int)sub[kb] : kb]; l = (int)ulist[u] >> lgG;
            x1 = ox[l]; y1 = oy[l]; z1 = oz[l]; m = mem[l];
        }
        Top2 r{-1e9f, -1e9f, -1, -1};                        // :116
        if (have && m != EMD_MEM_NONE) {
            // any two distinct targets bound the second-best value from below: the best two of the last bid, repriced
            const float v1 = emd_value(t4[m & 0xffffu], x1, y1, z1), v2 = emd_value(t4[m >> 16], x1, y1, z1);
            R = (3.0f - fminf(v1, v2)) + slack;
            if (!(R < 1e30f)) R = __builtin_inff();
            // PRICE FILTER.  A target can enter the best two only if its value reaches the bound the radius came from,
            // 3 - d - price >= 3 - R: d <= R - price.  In a crowded auction that holds for ~4 of the ~200-400 targets a
            // late bid walks (profiles/r04b_emd_price_sim.txt), so the walk tests d^2 <= (R - price)^2 -- ten plain
            // instructions, no square root; R carries `slack`, which covers the rounding of this test against the exact
            // value many times over -- and only the survivors get the exact value and the top-two update: a lane parks
            // its survivor and the wave evaluates the parked ones together when any lane finds a second one.  (The
            // best two over the survivors ARE the best two: both remembered targets survive, everything left out is
            // below both.)
            int pend = -1;
            emd_walk(x1, y1, z1, R, tl, T, Tb, [&](int k) {
                const bool pass = emd_may_matter(t4[k], x1, y1, z1, R);
                if (__builtin_amdgcn_ballot_w64(pass && pend >= 0)) {
                    if (pend >= 0) { top2_put(r, pend, emd_value(t4[pend], x1, y1, z1)); pend = -1; }
                }
                pend = pass ? k : pend;
            });
            if (pend >= 0) top2_put(r, pend, emd_value(t4[pend], x1, y1, z1));
        }
        top2_team(r, T);
        // first bid of a point (no memory yet): grow a ball until it holds the two best
        float Rt = (have && m == EMD_MEM_NONE) ? r0 : -1.0f;
        while (__builtin_amdgcn_ballot_w64(Rt > 0.0f)) {
            Top2 w{-1e9f, -1e9f, -1, -1};
            if (Rt > 0.0f) emd_walk(x1, y1, z1, Rt, tl, T, Tb, [&](int k) { top2_put(w, k, emd_value(t4[k], x1, y1, z1)); });
            top2_team(w, T);                                 // teams that are done fold empty results: theirs stays as it is
            if (Rt > 0.0f) {
                r = w;
                // every target outside the ball of radius 3 - second is strictly below the runner-up found so far
                const float need = (r.j >= 0 && r.s > -1e8f) ? (3.0f - r.s) + slack : __builtin_inff();
                if (need <= Rt || !(Rt < 1e30f)) Rt = -1.0f;                  // the ball already held them (or was everything)
                else Rt = (need < 1e30f) ? need : ((Rt < 64.0f * r0) ? Rt * 2.0f : __builtin_inff());
            }
        }
        // equal values at the top (lattices, duplicated targets): the lowest ORIGINAL index wins (:144-151 scans in
        // index order) -- the walk's order is not the index order, so the holders of the maximum are collected again
        const bool tie = have && r.i >= 0 && r.s == r.b;
        if (__builtin_amdgcn_ballot_w64(tie)) {
            int key = 0x7fffffff;
            if (tie) {
                const float Rw = (3.0f - r.b) + slack;
                emd_walk(x1, y1, z1, (Rw < 1e30f) ? Rw : __builtin_inff(), tl, T, Tb, [&](int k) {
                    if (emd_value(t4[k], x1, y1, z1) == r.b) key = min(key, ((int)orig[k] << 12) | k);
                });
            }
            key = emd_team_min(key, T);
            if (tie && key != 0x7fffffff) {
                const int ki = key & 0xfff;
                if (ki != r.i) { r.j = r.i; r.i = ki; }      // the displaced one holds the same value: still a valid memory
            }
        }
        if (have && tl == 0) {
            const bool two = r.i >= 0 && r.j >= 0 && r.i != r.j;
            mem[l] = two ? ((unsigned)r.i | ((unsigned)r.j << 16)) : EMD_MEM_NONE;
            if (!two) emd3_flat_off = 1;
            const float v = (r.b - r.s) + eps;                                          // :175-176
            const unsigned t = (unsigned)min(max(r.i, 0), n - 1);
            emd_st(bid_w + u, ((unsigned long long)((tag << 16) | t) << 32) | (unsigned)__float_as_int(v));
        }
    }

}

// -DEMD_TRACE (tools/emd_timeline.py): per workgroup and round, 100-MHz timestamps of the round's phases and the work
// counters, written behind the bid granules in the sample's workspace (8 words per (workgroup, round), 64 rounds)
#ifdef EMD_TRACE
#define EMD_TR(slot) do { if (tid == 0 && it < 64) trace[(g * 64 + it) * 8 + (slot)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EMD_TR(slot) do {} while (0)
#endif

__global__ __launch_bounds__(EMD_THREADS, 4) void emd_auction_team_kernel(const float* __restrict__ xyz1,
                                                                          const float* __restrict__ xyz2, int B, int n,
                                                                          int npad, int G, int lgG, float eps, int iters,
                                                                          float* __restrict__ dist, int32_t* assignment,
                                                                          float* wsf, unsigned* counters, int tnum, int tmax, int flat_min, int flat_lds, int flat_work) {
    extern __shared__ __attribute__((aligned(16))) float emd_lds[];
    const int nown = npad >> lgG;                                // points this workgroup bids for (local index i >> lgG)
    float4* t4 = reinterpret_cast<float4*>(emd_lds);                                               // SORTED by cell, then index
    unsigned long long* top_l = reinterpret_cast<unsigned long long*>(t4 + npad);                  // per target: (increment, ~bidder) max
    float* ox = reinterpret_cast<float*>(top_l + npad); float* oy = ox + nown; float* oz = oy + nown;   // own bidders
    unsigned* mem = reinterpret_cast<unsigned*>(oz + nown);                                        // best | second << 16 of each own bidder's last bid
    unsigned short* orig = reinterpret_cast<unsigned short*>(mem + nown);                          // sorted position -> target
    short* assign_l = reinterpret_cast<short*>(orig + npad);                                       // point -> sorted position | -1
    short* inv_l = assign_l + npad;                                                                // sorted position -> point | -1
    unsigned short* ulist = reinterpret_cast<unsigned short*>(inv_l + npad);                       // unassigned points, ascending
    unsigned short* ownu = ulist + npad;                                                           // list positions of the own ones
    unsigned short* const cell_start = emd3_cell_start;
    __shared__ float red[EMD_WAVES][6];
    EmdGrid3& grid = emd3_grid;
    unsigned* const wtot = emd3_wtot;
    __shared__ int gave_up;
    if (threadIdx.x == 0) { gave_up = 0; emd3_flat_off = 0; emd3_flat_tpb = 1.0e9f; }
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / G) * 8 + xcd, g = q % G;
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * n * 3;
    // workspace of the sample: two buffers (round parity) of n 8-byte bid granules
    unsigned long long* bids = reinterpret_cast<unsigned long long*>(wsf + (size_t)b * EMD_WS_PLANES * n);
    unsigned* counter = counters + 2 * b;
    unsigned passed = 0;
#ifdef EMD_TRACE
    unsigned* trace = reinterpret_cast<unsigned*>(wsf + (size_t)b * EMD_WS_PLANES * n + 4 * (size_t)n);
    unsigned& tr_evals = emd3_tr_evals; unsigned& tr_rows = emd3_tr_rows;
    { const int it = 0; if (tid == 0) trace[(g * 64 + 63) * 8 + 7] = (unsigned)__builtin_amdgcn_s_memrealtime(); (void)it; }
#endif
    int* cursor = reinterpret_cast<int*>(top_l);                 // init only: per-cell counters (4096 ints = top_l's 16 KB at n = 2048)
    // the grid is egx x 8 x 8 cells, x fastest: 32 along x when the counters have room, fewer for small clouds
    // (the host sends n < 128 to the other kernels: at least four)
    const int egx = min(ENC3, 2 * npad) / (EG3 * EG3), ncell = egx * EG3 * EG3;

    // ---- the grid: bounding box of the targets, counting sort by cell, rank inside a cell by original index
    {
        float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (int j = tid; j < n; j += EMD_THREADS)
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float c = p2[(size_t)j * 3 + a]; lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c); }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
            if (lane == 0) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
        }
        for (int c = tid; c < ncell; c += EMD_THREADS) cursor[c] = 0;
        __syncthreads();
        if (tid == 0) {
            float big = 0.0f;
            for (int a = 0; a < 3; ++a) {
                float l = red[0][a], h = red[0][3 + a];
                for (int w = 1; w < EMD_WAVES; ++w) { l = fminf(l, red[w][a]); h = fmaxf(h, red[w][3 + a]); }
                const float cells = a == 0 ? (float)egx : (float)EG3;
                grid.mn[a] = l;
                grid.sc[a] = h > l ? cells / (h - l) : 0.0f;      // a flat (or non-finite) extent: one layer of cells
                grid.cw[a] = h > l ? (h - l) / cells : 0.0f;
                big = fmaxf(big, fmaxf(fabsf(l), fabsf(h)));
            }
            grid.mg = 4.0e-6f * big + 1.0e-30f;                  // rounding of (x - mn) * sc and of mn + c * cw, with room
            grid.egx = egx;
        }
        __syncthreads();
        auto cell_of = [&](float x, float y, float z) {
            return (emd_cell3(z, grid.mn[2], grid.sc[2]) * EG3 + emd_cell3(y, grid.mn[1], grid.sc[1])) * egx
                   + emd_cell3(x, grid.mn[0], grid.sc[0], egx);
        };
        for (int j = tid; j < n; j += EMD_THREADS) atomicAdd(&cursor[cell_of(p2[(size_t)j * 3], p2[(size_t)j * 3 + 1], p2[(size_t)j * 3 + 2])], 1);
        __syncthreads();
        // exclusive scan of the cell counts: CPT cells per thread, waves, workgroup
        constexpr int CPT = ENC3 / EMD_THREADS;
        static_assert(ENC3 == CPT * EMD_THREADS && CPT >= 1, "whole cells per thread");
        int c4[CPT], sum4 = 0;
#pragma unroll
        for (int e = 0; e < CPT; ++e) {
            const int c = tid * CPT + e;
            c4[e] = c < ncell ? cursor[c] : 0;
            sum4 += c4[e];
        }
        const int incl = emd_wave_scan_incl(sum4);
        if (lane == 63) wtot[wave] = (unsigned)incl;
        __syncthreads();
        {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += (int)wtot[w];
            int run = before + incl - sum4;
#pragma unroll
            for (int e = 0; e < CPT; ++e) { cell_start[tid * CPT + e] = (unsigned short)run; run += c4[e]; }   // cells >= ncell: n
        }
        if (tid == 0) cell_start[ENC3] = (unsigned short)n;
        __syncthreads();
        for (int c = tid; c < ncell; c += EMD_THREADS) cursor[c] = cell_start[c];
        __syncthreads();
        for (int j = tid; j < n; j += EMD_THREADS)               // members of every cell, in whatever order the atomics give
            ulist[atomicAdd(&cursor[cell_of(p2[(size_t)j * 3], p2[(size_t)j * 3 + 1], p2[(size_t)j * 3 + 2])], 1)] = (unsigned short)j;
        __syncthreads();
        for (int j = tid; j < n; j += EMD_THREADS) {              // ... then by index: the same order in every workgroup
            const float x = p2[(size_t)j * 3], y = p2[(size_t)j * 3 + 1], z = p2[(size_t)j * 3 + 2];
            const int c = cell_of(x, y, z);
            const int s = cell_start[c], e = cell_start[c + 1];
            int rank = 0;
            for (int k = s; k < e; ++k) rank += (int)ulist[k] < j;
            t4[s + rank] = make_float4(x, y, z, 0.0f);
            orig[s + rank] = (unsigned short)j;
        }
        for (int j = n + tid; j < npad; j += EMD_THREADS) { t4[j] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff()); orig[j] = 0xffff; }
        for (int l = tid; l < nown; l += EMD_THREADS) {
            const int i = min((l << lgG) + g, n - 1);
            ox[l] = p1[(size_t)i * 3]; oy[l] = p1[(size_t)i * 3 + 1]; oz[l] = p1[(size_t)i * 3 + 2];
            mem[l] = EMD_MEM_NONE;
        }
        __syncthreads();                                         // cursor (aliases top_l) and the unsorted member lists are done with
        for (int j = tid; j < npad; j += EMD_THREADS) { assign_l[j] = -1; inv_l[j] = -1; top_l[j] = 0ull; }
    }
    for (int j = g * EMD_THREADS + tid; j < 2 * n; j += G * EMD_THREADS) emd_st(bids + j, 0ull);
    bool ok = emd_group_sync(counter, passed, G, &gave_up);       // the launch's only counter barrier: granule tags start at 0
    // first bid of a point (no memory yet): radius that holds ~6 targets of a cloud that fills its box; doubled until two are found
    const float r_first = 1.5f * cbrtf(fmaxf(grid.cw[0] * (float)egx, 1e-30f) * fmaxf(grid.cw[1] * EG3, 1e-30f) * fmaxf(grid.cw[2] * EG3, 1e-30f)
                                       * (6.0f / (4.0f * 3.14159265f)) / (float)n);
    const float r_first_floor = 0.03f * fmaxf(fmaxf(grid.cw[0] * (float)egx, grid.cw[1] * EG3), grid.cw[2] * EG3);
    const float r0 = fmaxf(r_first, r_first_floor) > 0.0f ? fmaxf(r_first, r_first_floor) : 1.0f;
    const float slack = 1.0e-5f + 4.0f * grid.mg;

    int team_run = 0;                                    // rounds since the last balanced one
    for (int it = 0; ok && it < iters; ++it) {
        const bool last = it == iters - 1;
        unsigned long long* bid_w = bids + (size_t)(it & 1) * n;
        const unsigned tag = (unsigned)(it % 65535) + 1u;        // never 0, differs from the tag two rounds ago
        EMD_TR(0);
#ifdef EMD_TRACE
        if (tid == 0) { tr_evals = 0; tr_rows = 0; }
#endif
        // ---- unassigned points in ascending order (every copy builds the same list) and the own ones among them:
        //      thread t looks at points t and t + 1024 (same residue mod G); four 8-bit counters in one DPP scan
        const bool mine = (tid & (G - 1)) == g;
        const int c0 = (tid < n && assign_l[tid] == -1) ? 1 : 0;
        const int c1 = (tid + EMD_THREADS < n && assign_l[tid + EMD_THREADS] == -1) ? 1 : 0;
        const int pk = c0 | (c1 << 8) | ((mine ? c0 : 0) << 16) | ((mine ? c1 : 0) << 24);
        const int sc = emd_wave_scan_incl(pk);
        if (lane == 63) wtot[wave] = (unsigned)sc;
        __syncthreads();
        // the 16 wave totals -> this wave's offsets: lanes 0..15 scan them on the VALU (a 16-step loop over LDS words in
        // every lane was ~130 instructions of every wave of every round).  16-bit fields: (c0 | own c0 << 16), (c1 | own c1 << 16)
        unsigned befA, totA, befB, totB;
        {
            const unsigned t = wtot[lane & (EMD_WAVES - 1)];
            int a = (int)(t & 0x00ff00ffu), bb = (int)((t >> 8) & 0x00ff00ffu);
            const int a0 = a, b0 = bb;
#define EMD_RSTEP(v, ctrl) v += __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false)
            EMD_RSTEP(a, 0x111); EMD_RSTEP(a, 0x112); EMD_RSTEP(a, 0x114); EMD_RSTEP(a, 0x118);      // row_shr 1, 2, 4, 8: inclusive
            EMD_RSTEP(bb, 0x111); EMD_RSTEP(bb, 0x112); EMD_RSTEP(bb, 0x114); EMD_RSTEP(bb, 0x118);
#undef EMD_RSTEP
            static_assert(EMD_WAVES == 16, "one DPP row holds the wave totals");
            totA = (unsigned)__builtin_amdgcn_readlane(a, 15); totB = (unsigned)__builtin_amdgcn_readlane(bb, 15);
            befA = (unsigned)__builtin_amdgcn_readlane(a - a0, wave); befB = (unsigned)__builtin_amdgcn_readlane(bb - b0, wave);
        }
        const int U = (int)((totA & 0xffffu) + (totB & 0xffffu));
        const int Uown = (int)((totA >> 16) + (totB >> 16));
        if (U == 0) break;                                       // every copy agrees
        {
            const unsigned ex = (unsigned)(sc - pk);             // exclusive prefix inside the wave
            const int u0 = (int)((befA & 0xffffu) + (ex & 0xffu));
            const int u1 = (int)((totA & 0xffffu) + (befB & 0xffffu) + ((ex >> 8) & 0xffu));
            if (c0) ulist[u0] = (unsigned short)tid;
            if (c1) ulist[u1] = (unsigned short)(tid + EMD_THREADS);
            if (mine) {
                if (c0) ownu[(befA >> 16) + ((ex >> 16) & 0xffu)] = (unsigned short)u0;
                if (c1) ownu[(totA >> 16) + (befB >> 16) + ((ex >> 24) & 0xffu)] = (unsigned short)u1;
            }
        }
        __syncthreads();

        EMD_TR(1);
        // ---- Bid (:95-179), pruned: teams of T lanes, one own bidder each
        // T lanes per bidder: the own bidders spread over ~tnum lanes, not over all 1024 -- a team's lanes repeat the row
        // set-up, and the rounds are bound by instruction issue (two workgroups share a CU), not by idle lanes
        // rounds with many bidders: the balanced form (emd_flat_bid above)
#ifdef EMD_TRACE
        unsigned* ftrace = (tid == 0 && it < 64) ? reinterpret_cast<unsigned*>(wsf + (size_t)b * EMD_WS_PLANES * n + 8 * (size_t)n) + (g * 64 + it) * 4 : nullptr;
#else
        unsigned* ftrace = nullptr;
#endif
        // Which form: by default the balanced one from the second round on (the first bids have no memory: their ball has
        // to grow, the teams do that).  The switch that remains for experiments: at least flat_min own bidders, and own
        // bidders x targets per bid of the last balanced round >= flat_work (a run of team rounds is interrupted every
        // eighth round to measure again).
        const bool flat = flat_min > 0 && it > 0 && Uown >= flat_min && flat_lds && !emd3_flat_off &&
                          ((float)Uown * emd3_flat_tpb >= (float)flat_work || team_run >= 7);
        team_run = flat ? 0 : team_run + 1;
        if (flat) emd_flat_bid(npad, lgG, Uown, n, eps, slack, tag, bid_w, ftrace, -1.0f);
        else {
            // the first bids: the balanced form with the radius the teams would start from, the teams for whoever it leaves
            int left = -1;
            if (it == 0 && flat_min > 0 && flat_lds && Uown >= flat_min && Uown <= EMD_FLAT_SLOTS)
                left = emd_flat_bid(npad, lgG, Uown, n, eps, slack, tag, bid_w, ftrace, r0);
            if (left < 0) emd_team_bid(npad, lgG, Uown, n, eps, slack, r0, tag, bid_w, tnum, tmax);
            else if (left > 0) emd_team_bid(npad, lgG, left, n, eps, slack, r0, tag, bid_w, tnum, tmax, emd3_flat_frow(npad, lgG));
        }

        EMD_TR(2);
#ifdef EMD_TRACE
        if (tid == 0 && it < 64) { trace[(g * 64 + it) * 8 + 5] = (unsigned)U | ((unsigned)Uown << 16); }
#endif
        // ---- gather the round's granules (one lane per bidder), GetMax (:181-194) on this copy
        int gi[2], gt[2]; unsigned long long gk[2]; float gv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = tid + h * EMD_THREADS;
            gi[h] = -1; gt[h] = 0; gk[h] = 0ull; gv[h] = 0.0f;
            if (u < U) {
                unsigned long long e = emd_ld(bid_w + u);
                unsigned spins = 0;
                while ((unsigned)(e >> 48) != tag) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > EMD_SPIN_LIMIT ||
                        ((spins & 255u) == 0 && __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tell the others
                        gave_up = 1;
                        break;
                    }
                    e = emd_ld(bid_w + u);
                }
                gi[h] = ulist[u];
                gt[h] = min((int)((e >> 32) & 0xffffu), n - 1);
                gv[h] = __int_as_float((int)(unsigned)e);
                gk[h] = ((e & 0xffffffffull) << 32) | (unsigned)(0x7fffffff - gi[h]);       // v >= 0: its bits order like the value
                if (!last) atomicMax(&top_l[gt[h]], gk[h]);
            }
        }
        __syncthreads();
        EMD_TR(3);
        if (gave_up) { ok = false; break; }
        // ---- Assign (:196-215) of ALL bidders on this workgroup's copy
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (gi[h] < 0) continue;
            const int i = gi[h], t = gt[h];
            if (last) { assign_l[i] = (short)t; continue; }
            if (top_l[t] != gk[h]) continue;
            const int prev = inv_l[t];
            if (prev != -1) assign_l[prev] = -1;
            inv_l[t] = (short)i;
            assign_l[i] = (short)t;
            t4[t].w = t4[t].w + gv[h];                          // :211
            top_l[t] = 0ull;                                    // :212 (losers read either their winner's key or 0: both != theirs)
        }
        __syncthreads();
        EMD_TR(4);
#ifdef EMD_TRACE
        if (tid == 0 && it < 64) { trace[(g * 64 + it) * 8 + 6] = tr_evals; trace[(g * 64 + it) * 8 + 7] = tr_rows; }
#endif
    }

    if (!ok) {                                                  // a partner never delivered: no result for this sample
        for (int l = tid; l < nown; l += EMD_THREADS) {
            const int i = (l << lgG) + g;
            if (i < n) { dist[(size_t)b * n + i] = __builtin_nanf(""); assignment[(size_t)b * n + i] = -1; }
        }
        return;
    }
    for (int l = tid; l < nown; l += EMD_THREADS) {               // CalcDist :217-226 + the assignment itself, own points
        const int i = (l << lgG) + g;
        if (i >= n) continue;
        const int t = assign_l[i];
        if (t < 0 || t >= n) { assignment[(size_t)b * n + i] = -1; dist[(size_t)b * n + i] = __builtin_nanf(""); continue; }
        assignment[(size_t)b * n + i] = (int)orig[t];
        const float4 y = t4[t];
        const float dx = ox[l] - y.x, dy = oy[l] - y.y, dz = oz[l] - y.z;
        dist[(size_t)b * n + i] = ((dx * dx) + (dy * dy)) + (dz * dz);
    }
}

// NmDistanceGradKernel :284-300: grad_xyz1 = (2 g) (x1 - x2[assignment]); xyz2 gets no gradient
__global__ __launch_bounds__(256) void emd_bwd_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                      const float* __restrict__ grad_dist,
                                                      const int32_t* __restrict__ assignment, int n, int total,
                                                      float* __restrict__ grad_xyz1) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int b = e / n, t = assignment[e];
    if (t < 0 || t >= n) {          // a point the forward left unassigned (its distance is NaN): the gradient says so too,
        const float nan = __builtin_nanf("");      // and nothing is read through the index
        grad_xyz1[(size_t)e * 3] = nan; grad_xyz1[(size_t)e * 3 + 1] = nan; grad_xyz1[(size_t)e * 3 + 2] = nan;
        return;
    }
    const float* a = xyz1 + (size_t)e * 3;
    const float* c = xyz2 + ((size_t)b * n + t) * 3;
    const float g = grad_dist[e] * 2.0f;
    grad_xyz1[(size_t)e * 3] = g * (a[0] - c[0]);
    grad_xyz1[(size_t)e * 3 + 1] = g * (a[1] - c[1]);
    grad_xyz1[(size_t)e * 3 + 2] = g * (a[2] - c[2]);
}

}  // namespace vpn

using namespace vpn;

// Largest group size G (power of two) such that the whole grid is resident: workgroups per CU from the occupancy
// query of THIS kernel on the CURRENT device (asked every call: nothing is cached across devices) times its CU
// count.  max_group caps it (1 = no inter-workgroup barrier at all).
static int emd_group_size_of(const void* kernel, size_t dyn_lds, int B, int n, int max_group) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, EMD_THREADS, dyn_lds) != hipSuccess || per_cu <= 0)
        return 1;
    const long long slots = (long long)cus * per_cu;
    const int padded = (B + 7) / 8 * 8;
    const int cap = max_group > 0 && max_group < EMD_MAX_GROUP ? max_group : EMD_MAX_GROUP;
    int G = 1;
    while (G * 2 <= cap && (long long)padded * G * 2 <= slots && G * 2 * EMD_WAVES * 4 <= n) G *= 2;
    return G;
}

static int emd_group_size(int B, int n, int max_group) {
    return emd_group_size_of(reinterpret_cast<const void*>(emd_auction_kernel), 0, B, n, max_group);
}

// VPN_EMD_STREAMING=1: the streaming kernel also for one-tile problems; VPN_EMD_NOGRID=1: the unpruned replicated-state
// kernel also for n <= 2048 (cross-checks of the three kernels in the tests)
static bool emd_env_flag(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}
static bool emd_force_streaming() { return emd_env_flag("VPN_EMD_STREAMING"); }
// The group size G is bounded by hipOccupancyMaxActiveBlocksPerMultiprocessor x #CU on every call, so the grid of a PLAIN
// launch is resident as a whole unless something else holds the CUs (then the bounded spin gives up: NaN / -1, a loud
// failure).  VPN_EMD_COOP_LAUNCH=1 launches cooperatively instead, which adds the runtime's own residency check -- off by
// default since round 4: in a process that has captured a HIP graph, ONE cooperative launch makes every later dispatch of
// the process ~50 us slower (profiles/r04_coop_launch_side_effect.txt: the C5 step 1.90 -> 2.55 ms, every kernel of it
// +45..65 us, eager and replayed), and it costs ~40 us of host time per call.
static bool emd_coop_launch() { return emd_env_flag("VPN_EMD_COOP_LAUNCH"); }

static size_t emd_state_bytes(int B, int n) { return (size_t)B * n * EMD_WS_PLANES * sizeof(float); }

extern "C" size_t vpn_emd_workspace(int B, int n) {
    if (B <= 0 || n <= 0) return 0;
    return emd_state_bytes(B, n) + ((size_t)2 * B * sizeof(unsigned) + 7) / 8 * 8;
}

extern "C" int vpn_emd_fwd(const float* xyz1, const float* xyz2, int B, int n, float eps, int iters, float* dist,
                           int32_t* assignment, void* workspace, int max_group, void* stream) {
    if (!xyz1 || !xyz2 || !dist || !assignment || !workspace || B < 0 || n < 0 || iters < 1 || !(eps >= 0.0f))
        return VPN_E_BADARG;
    if ((long long)B * n * 3 > 0x7fffffffLL) return VPN_E_TOOBIG;
    if (((uintptr_t)workspace & 7) != 0) return VPN_E_BADARG;
    if (B == 0 || n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    unsigned* counters = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + emd_state_bytes(B, n));
    if (hipMemsetAsync(counters, 0, (size_t)2 * B * sizeof(unsigned), s) != hipSuccess) return (int)hipGetLastError();
    float* wsf = (float*)workspace;
    if (n <= EMD_GRID_MAX && n >= 128 && !emd_force_streaming() && !emd_env_flag("VPN_EMD_NOGRID") && !emd_env_flag("VPN_EMD_GRID1")) {
        // the training call (n = SAMPLE_NUM * VP_NUM = 2048): pruned scan, static ownership, granule exchange
        const int npad = (n + 63) / 64 * 64;
        const void* kern = reinterpret_cast<const void*>(emd_auction_team_kernel);
        const int flat_lds = npad >= 1024 ? 1 : 0;               // the balanced form's lists (small clouds never need it)
        auto lds_of = [&](int G) { return (size_t)npad * 32 + (size_t)(npad / G) * 18 + (flat_lds ? EMD_FLAT_BYTES : 0); };
        static size_t raised = 0;
        if (lds_of(1) > raised) {
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(1));
            if (e != hipSuccess) return (int)e;
            raised = lds_of(1);
        }
        // largest G (power of two, <= 16, <= max_group) whose whole grid is resident: workgroups per CU from the
        // occupancy query of THIS kernel with THAT group size's LDS, times the CU count of the current device
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        const int padded = (B + 7) / 8 * 8;
        const int cap = max_group > 0 && max_group < EMD_MAX_GROUP ? max_group : EMD_MAX_GROUP;
        int G = 1, lgG = 0;
        for (int cand = EMD_MAX_GROUP, lg = 4; cand > 1; cand >>= 1, --lg) {
            if (cand > cap || cand * EMD_WAVES * 4 > n || npad % cand != 0) continue;
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, EMD_THREADS, lds_of(cand)) != hipSuccess || per_cu <= 0) continue;
            if ((long long)padded * cand <= (long long)cus * per_cu) { G = cand; lgG = lg; break; }
        }
        const unsigned lds = (unsigned)lds_of(G);
        const bool coop = G > 1 && emd_coop_launch();
        int tnum = 1024, tmax = 16;                              // lanes the own bidders of a round are spread over; largest team
        if (const char* e = getenv("VPN_EMD_TNUM")) tnum = atoi(e) > 0 ? atoi(e) : tnum;
        if (const char* e = getenv("VPN_EMD_TMAX")) tmax = atoi(e) > 0 ? atoi(e) : tmax;
        if (tmax > 64) tmax = 64;
        // the balanced form bids every round after the first (measured: with 1 to 64 lanes per row it beats the teams in sparse
        // rounds too); VPN_EMD_FLAT_MIN own bidders / VPN_EMD_FLAT_WORK targets in a round's rows move the switch (0 / -: teams only)
        int flat_min = 1, flat_work = 0;
        if (const char* e = getenv("VPN_EMD_FLAT_MIN")) flat_min = atoi(e);
        if (const char* e = getenv("VPN_EMD_FLAT_WORK")) flat_work = atoi(e);
        if (coop) {
            void* args[] = {(void*)&xyz1, (void*)&xyz2, (void*)&B, (void*)&n, (void*)&npad, (void*)&G, (void*)&lgG, (void*)&eps, (void*)&iters,
                            (void*)&dist, (void*)&assignment, (void*)&wsf, (void*)&counters, (void*)&tnum, (void*)&tmax, (void*)&flat_min, (void*)&flat_lds, (void*)&flat_work};
            vpn::prof_begin("emd_auction_team_kernel", s);
            const hipError_t e = hipLaunchCooperativeKernel(kern, dim3(padded * G), dim3(EMD_THREADS), args, lds, s);
            vpn::prof_end(s);
            if (e == hipSuccess) return 0;
            (void)hipGetLastError();
            if (e != hipErrorCooperativeLaunchTooLarge && e != hipErrorNotSupported && e != hipErrorInvalidConfiguration) return (int)e;
            G = 1; lgG = 0;
        }
        VPN_LAUNCH(emd_auction_team_kernel, dim3(padded * G), dim3(EMD_THREADS), (unsigned)lds_of(G), s, xyz1, xyz2, B, n, npad, G, lgG, eps,
                   iters, dist, assignment, wsf, counters, tnum, tmax, flat_min, flat_lds, flat_work);
        VPN_LAUNCH_CHECK();
        return 0;
    }
    if (n <= EMD_TILE && !emd_force_streaming()) {
        // one-tile problem (every training call): replicated state, one group barrier per round
        int npad = (n + 64 * EMD_UNROLL - 1) / (64 * EMD_UNROLL) * (64 * EMD_UNROLL);
        const bool grid = n <= EMD_GRID_MAX && n >= 64 && !emd_env_flag("VPN_EMD_NOGRID");     // pruned Bid scan
        const void* kern = grid ? reinterpret_cast<const void*>(emd_auction_grid_kernel)
                                : reinterpret_cast<const void*>(emd_auction_local_kernel);
        const size_t lds = (size_t)npad * (grid ? 12 : 7) * sizeof(float);
        static size_t raised[2] = {0, 0};              // largest dynamic LDS size each kernel has been allowed so far
        if (lds > raised[grid]) {
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            raised[grid] = lds;
        }
        int G = emd_group_size_of(kern, lds, B, n, max_group);
        if (G > 1 && emd_coop_launch()) {
            void* args[] = {(void*)&xyz1, (void*)&xyz2, (void*)&B, (void*)&n, (void*)&npad, (void*)&G, (void*)&eps, (void*)&iters,
                            (void*)&dist, (void*)&assignment, (void*)&wsf, (void*)&counters};
            vpn::prof_begin(grid ? "emd_auction_grid_kernel" : "emd_auction_local_kernel", s);
            const hipError_t e = hipLaunchCooperativeKernel(kern, dim3((B + 7) / 8 * 8 * G), dim3(EMD_THREADS), args, (unsigned)lds, s);
            vpn::prof_end(s);
            if (e == hipSuccess) return 0;
            (void)hipGetLastError();
            if (e != hipErrorCooperativeLaunchTooLarge && e != hipErrorNotSupported && e != hipErrorInvalidConfiguration) return (int)e;
            G = 1;
        }
        if (grid)
            VPN_LAUNCH(emd_auction_grid_kernel, dim3((B + 7) / 8 * 8 * G), dim3(EMD_THREADS), lds, s, xyz1, xyz2, B, n, npad, G, eps,
                       iters, dist, assignment, wsf, counters);
        else
            VPN_LAUNCH(emd_auction_local_kernel, dim3((B + 7) / 8 * 8 * G), dim3(EMD_THREADS), lds, s, xyz1, xyz2, B, n, npad, G, eps,
                       iters, dist, assignment, wsf, counters);
        VPN_LAUNCH_CHECK();
        return 0;
    }
    int G = emd_group_size(B, n, max_group);
    if (G > 1 && emd_coop_launch()) {
        // the G workgroups of a sample synchronise with each other: a COOPERATIVE launch makes the runtime check that
        // the whole grid can be resident at once; if it says no, fall back to one workgroup per sample
        void* args[] = {(void*)&xyz1, (void*)&xyz2, (void*)&B, (void*)&n, (void*)&G, (void*)&eps, (void*)&iters,
                        (void*)&dist, (void*)&assignment, (void*)&wsf, (void*)&counters};
        vpn::prof_begin("emd_auction_kernel", s);
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(emd_auction_kernel),
                                                        dim3((B + 7) / 8 * 8 * G), dim3(EMD_THREADS), args, 0, s);
        vpn::prof_end(s);
        if (e == hipSuccess) return 0;
        (void)hipGetLastError();
        if (e != hipErrorCooperativeLaunchTooLarge && e != hipErrorNotSupported && e != hipErrorInvalidConfiguration) return (int)e;
        G = 1;
    }
    VPN_LAUNCH(emd_auction_kernel, dim3((B + 7) / 8 * 8 * G), dim3(EMD_THREADS), 0, s, xyz1, xyz2, B, n, G, eps, iters,
               dist, assignment, wsf, counters);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_emd_bwd(const float* xyz1, const float* xyz2, const float* grad_dist, const int32_t* assignment,
                           int B, int n, float* grad_xyz1, void* stream) {
    if (!xyz1 || !xyz2 || !grad_dist || !assignment || !grad_xyz1 || B < 0 || n < 0) return VPN_E_BADARG;
    if ((long long)B * n * 3 > 0x7fffffffLL) return VPN_E_TOOBIG;
    if (B == 0 || n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int total = B * n;
    VPN_LAUNCH(emd_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, xyz1, xyz2, grad_dist, assignment, n, total,
               grad_xyz1);
    VPN_LAUNCH_CHECK();
    return 0;
}

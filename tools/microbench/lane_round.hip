// What does one iteration of the light items' lane rounds cost, and which coding of the SAME arithmetic is fastest?
// (round 6: the longest light item -- ~360 iterations at 200-250 ns -- is the send launch's critical path.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I pcc-rl_amd/csrc tools/microbench/lane_round.hip -o lane_round
// 1 024 wavefronts (256 workgroups of 4: one wavefront per SIMD, like the light workgroups), a lane per env, every env
// ~300-360 packets of one monitor interval on a link of the ICML'19 ranges, records to a tier-0 ring of its own (24 KB apart).
// Variants (all but 9 must leave bit-identical state and records -- checked against variant 0):
//   0  the product's loop as it is (pcc_send_item.h: philox4x32_10 inline asm, link_send, offset by ?:)
//   1  branch-free packet: masks instead of control flow, the tail-drop test folded with the loss bit, selects off the chain
//   2  1 + Philox of the NEXT block computed in the same basic block (compiler-visible 64-bit multiplies)
//   3  2 with two blocks (8 packets) per loop body
//   4  2 with the inline-asm multiply kept
//   5  2 without the record stores; 6  0 without the record stores (timing only)
//   9  1 without Philox (loss bits from a counter: WRONG results, a floor for "Philox off the lane round")
// arg 3: probe wavefronts (1 024 = one per SIMD in workgroups of four; fewer: workgroups of one wavefront, e.g. 256 = one per compute unit, 32 = one per 8)
// arg 2: neighbours per SIMD running variant 0 on envs of their own (0..3).
// arg 5: lanes with packets per wavefront (64; 32 / 16: smaller lane-round items).
#include "pcc_dev.h"
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct EnvP {
    double dl, lr, maxq, ebw, q, tu, t, gap, end;
    uint32_t gid, episode, mi, a, d, pad[3];
};
struct EnvOut {
    double q, tu, t;
    uint32_t a, d;
    double sum_x, sum_y;   // (filled by the host from the rings)
};
constexpr uint32_t kCap = 512;
constexpr size_t kRingBytes = (size_t)3 * kCap * 16;

namespace {

__device__ __forceinline__ void philox_c(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)c0 * 0xD2511F53u, p1 = (uint64_t)c2 * 0xCD9E8D57u;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Lane {
    double dl, maxq, ebw, gap, end, q, tu, t;
    uint32_t thr, a, d, gid, episode, mi;
    bool always;
    char *base;
};

// ---- variant 0: the product's loop
template <bool STORE = true>
__device__ __forceinline__ void rounds_v0(Lane &L, uint32_t key0, uint32_t key1, uint32_t round_packets) {
    const double dl = L.dl, maxq = L.maxq, ebw = L.ebw, gap = L.gap, end = L.end;
    const uint32_t thr = L.thr, gid = L.gid, episode = L.episode, mi = L.mi;
    const bool always = L.always;
    char *base = L.base;
    const uint32_t mask_b = (kCap - 1u) << 4, dmask_b = (2u * kCap - 1u) << 4, cap_b = kCap << 4;
    double q = L.q, tu = L.tu, t = L.t;
    uint32_t a = L.a, d = L.d, blk = 0;
    bool active = t < end;
    for (;;) {
        if (active) {
            const double ahead = (end - t) / gap - 2.0;
            uint32_t safe4 = ahead >= 4.0 ? (uint32_t)fmin(ahead, (double)round_packets) >> 2 : 0u;
            uint32_t budget4 = round_packets / 4 - safe4;
            for (; safe4; safe4--) {
                uint32_t w[4];
                philox4x32_10(blk, mi, episode, gid, key0, key1, w);
                blk++;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bool dropped;
                    const double2 rec = link_send(t, always || w[k] < thr, dl, maxq, ebw, q, tu, dropped);
                    const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                    if (STORE) st_rec(reinterpret_cast<double2 *>(base + off), rec);
                    else asm volatile("" :: "v"(rec.x), "v"(rec.y), "v"(off));
                    a += dropped ? 0u : 1u;
                    d += dropped ? 1u : 0u;
                    t += gap;
                }
            }
            for (; budget4 && t < end; budget4--) {
                uint32_t w[4];
                philox4x32_10(blk, mi, episode, gid, key0, key1, w);
                blk++;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k > 0 && !(t < end)) break;
                    bool dropped;
                    const double2 rec = link_send(t, always || w[k] < thr, dl, maxq, ebw, q, tu, dropped);
                    const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                    if (STORE) st_rec(reinterpret_cast<double2 *>(base + off), rec);
                    else asm volatile("" :: "v"(rec.x), "v"(rec.y), "v"(off));
                    a += dropped ? 0u : 1u;
                    d += dropped ? 1u : 0u;
                    t += gap;
                }
            }
            active = t < end;
        }
        if (!__ballot(active)) break;
    }
    L.q = q; L.tu = tu; L.t = t; L.a = a; L.d = d;
}

// ---- the branch-free packet: the same operations of ns:66-84 in the same order, other selects
//   qcur  = max(0, q - (t - tu))                              ns:66-67
//   grown = qcur + 1/bw  (= 1/bw + qcur, the operands of ns:79 and ns:82 commute)
//   dropped = lost at random || grown > maxq                  ns:73, 79: ONE compare against (lost ? a negative number : maxq)
//   q'    = dropped ? (lost ? q : qcur) : grown               the inner select does not wait for the compare
//   tu'   = lost ? tu : t
// a4 / d4: the ring indices times 16.
struct PacketState {
    double q, tu, t;
    uint32_t a4, d4;
};
__device__ __forceinline__ void packet_bf(PacketState &S, const bool lost, const double dl, const double maxq, const double ebw, const double gap,
                                          char *base, const uint32_t mask_b, const uint32_t dmask_b, const uint32_t cap_b, const bool store = true) {
    const double t = S.t;
    const double qcur = max0(S.q - (t - S.tu));
    const double grown = qcur + ebw;
    const double lat0 = dl + qcur;
    // (lost: compare against -|maxq| - 1.xxx: any negative number is below grown >= 1/bw > 0)
    const double lim = __hiloint2double(lost ? (int)0xBFF00000u : __double2hiint(maxq), __double2loint(maxq));
    const bool dropped = grown > lim;
    const double keep = lost ? S.q : qcur;
    S.q = dropped ? keep : grown;
    S.tu = lost ? S.tu : t;
    double2 rec;
    rec.x = t + lat0;
    rec.y = lat0;
    const uint32_t off_a = S.a4 & mask_b, off_d = cap_b + (S.d4 & dmask_b);
    if (store) st_rec(reinterpret_cast<double2 *>(base + (dropped ? off_d : off_a)), rec);
    else asm volatile("" :: "v"(rec.x), "v"(rec.y), "v"(dropped ? off_d : off_a));
    const uint32_t inc = dropped ? 16u : 0u;
    S.d4 += inc;
    S.a4 += 16u - inc;
    S.t = t + gap;
}

template <int V>
__device__ __forceinline__ void philox_v(uint32_t blk, uint32_t mi, uint32_t episode, uint32_t gid, uint32_t key0, uint32_t key1, uint32_t (&w)[4]) {
    if (V == 9) { w[0] = blk * 2654435761u; w[1] = w[0] ^ gid; w[2] = w[1] * 40503u; w[3] = w[2] ^ mi; }
    else if (V == 1 || V == 4) philox4x32_10(blk, mi, episode, gid, key0, key1, w);
    else philox_c(blk, mi, episode, gid, key0, key1, w);
}

template <int V>
__device__ __forceinline__ void rounds_bf(Lane &L, uint32_t key0, uint32_t key1, uint32_t round_packets) {
    const double dl = L.dl, maxq = L.maxq, ebw = L.ebw, gap = L.gap, end = L.end;
    const uint32_t thr = L.thr, gid = L.gid, episode = L.episode, mi = L.mi;
    const bool always = L.always;
    char *base = L.base;
    const uint32_t mask_b = (kCap - 1u) << 4, dmask_b = (2u * kCap - 1u) << 4, cap_b = kCap << 4;
    PacketState S;
    S.q = L.q; S.tu = L.tu; S.t = L.t; S.a4 = L.a << 4; S.d4 = L.d << 4;
    constexpr bool kPipe = V == 2 || V == 3 || V == 4 || V == 5;
    constexpr bool kStore = V != 5;
    constexpr int kBlocks = V == 3 ? 2 : 1;   // Philox blocks per loop body
    uint32_t blk = 0;
    bool active = S.t < end;
    for (;;) {
        if (active) {
            const double ahead = (end - S.t) / gap - 2.0;
            uint32_t safe4 = ahead >= 4.0 ? (uint32_t)fmin(ahead, (double)round_packets) >> 2 : 0u;
            uint32_t budget4 = round_packets / 4 - safe4;
            if (kPipe) {
                uint32_t w[kBlocks][4];
#pragma unroll
                for (int b = 0; b < kBlocks; b++) philox_v<V>(blk + b, mi, episode, gid, key0, key1, w[b]);
                for (; safe4 >= (uint32_t)kBlocks; safe4 -= kBlocks) {
                    uint32_t wn[kBlocks][4];
#pragma unroll
                    for (int b = 0; b < kBlocks; b++) philox_v<V>(blk + kBlocks + b, mi, episode, gid, key0, key1, wn[b]);
#pragma unroll
                    for (int b = 0; b < kBlocks; b++)
#pragma unroll
                        for (int k = 0; k < 4; k++) packet_bf(S, always || w[b][k] < thr, dl, maxq, ebw, gap, base, mask_b, dmask_b, cap_b, kStore);
                    blk += kBlocks;
#pragma unroll
                    for (int b = 0; b < kBlocks; b++)
#pragma unroll
                        for (int k = 0; k < 4; k++) w[b][k] = wn[b][k];
                }
                budget4 += safe4;   // (a block left over when two go to a body)
                for (; budget4 && S.t < end; budget4--) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (k > 0 && !(S.t < end)) break;
                        packet_bf(S, always || w[0][k] < thr, dl, maxq, ebw, gap, base, mask_b, dmask_b, cap_b, kStore);
                    }
                    blk++;
                    philox_v<V>(blk, mi, episode, gid, key0, key1, w[0]);
                }
            } else {
                for (; safe4; safe4--) {
                    uint32_t w[4];
                    philox_v<V>(blk, mi, episode, gid, key0, key1, w);
                    blk++;
#pragma unroll
                    for (int k = 0; k < 4; k++) packet_bf(S, always || w[k] < thr, dl, maxq, ebw, gap, base, mask_b, dmask_b, cap_b, kStore);
                }
                for (; budget4 && S.t < end; budget4--) {
                    uint32_t w[4];
                    philox_v<V>(blk, mi, episode, gid, key0, key1, w);
                    blk++;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (k > 0 && !(S.t < end)) break;
                        packet_bf(S, always || w[k] < thr, dl, maxq, ebw, gap, base, mask_b, dmask_b, cap_b, kStore);
                    }
                }
            }
            active = S.t < end;
        }
        if (!__ballot(active)) break;
    }
    L.q = S.q; L.tu = S.tu; L.t = S.t; L.a = S.a4 >> 4; L.d = S.d4 >> 4;
}


// ---- variant 7: variant 2's packet, the ACCEPTED records of a Philox block transposed through LDS and written 16 envs x (up to) 64
// consecutive ring bytes per store instruction; a dropped record leaves at once (predicated scattered store).  The block loop is
// wave-uniform (lanes that have run out publish an empty row).
struct StageLds {
    double2 row[64][5];   // [lane][j-th accepted record of the block; 4 = the slot a dropped record's copy goes to]
    uint4 pub[64];        // [lane] ring base (2 words), byte offset of the block's first accepted record before masking, mask | count
};
__device__ __forceinline__ void packet_t(PacketState &S, const bool lost, const double dl, const double maxq, const double ebw, const double gap,
                                         char *base, const uint32_t dmask_b, const uint32_t cap_b, double2 *row, uint32_t &jk) {
    const double t = S.t;
    const double qcur = max0(S.q - (t - S.tu));
    const double grown = qcur + ebw;
    const double lat0 = dl + qcur;
    const double lim = __hiloint2double(lost ? (int)0xBFF00000u : __double2hiint(maxq), __double2loint(maxq));
    const bool dropped = grown > lim;
    const double keep = lost ? S.q : qcur;
    S.q = dropped ? keep : grown;
    S.tu = lost ? S.tu : t;
    double2 rec;
    rec.x = t + lat0;
    rec.y = lat0;
    row[dropped ? 4u : jk] = rec;
    if (dropped) st_rec(reinterpret_cast<double2 *>(base + cap_b + (S.d4 & dmask_b)), rec);
    const uint32_t inc = dropped ? 16u : 0u;
    jk += dropped ? 0u : 1u;
    S.d4 += inc;
    S.a4 += 16u - inc;
    S.t = t + gap;
}
__device__ __forceinline__ void flush_block(StageLds &X, const uint32_t lane, char *base, const uint32_t a_old4, const uint32_t mask_b, const uint32_t jk) {
    X.pub[lane] = make_uint4((uint32_t)(uintptr_t)base, (uint32_t)((uintptr_t)base >> 32), a_old4, mask_b | jk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t r = lane & 3u;
#pragma unroll
    for (uint32_t g = 0; g < 4u; g++) {
        const uint32_t sl = 16u * g + (lane >> 2);
        const uint4 P = X.pub[sl];
        const double2 R = X.row[sl][r];
        if (r < (P.w & 7u)) {
            char *b = reinterpret_cast<char *>((uintptr_t)P.x | ((uintptr_t)P.y << 32));
            st_rec(reinterpret_cast<double2 *>(b + ((P.z + r * 16u) & (P.w & ~15u))), R);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void rounds_t(Lane &L, uint32_t key0, uint32_t key1, uint32_t round_packets, StageLds &X, const uint32_t lane) {
    const double dl = L.dl, maxq = L.maxq, ebw = L.ebw, gap = L.gap, end = L.end;
    const uint32_t thr = L.thr, gid = L.gid, episode = L.episode, mi = L.mi;
    const bool always = L.always;
    char *base = L.base;
    const uint32_t mask_b = (kCap - 1u) << 4, dmask_b = (2u * kCap - 1u) << 4, cap_b = kCap << 4;
    PacketState S;
    S.q = L.q; S.tu = L.tu; S.t = L.t; S.a4 = L.a << 4; S.d4 = L.d << 4;
    double2 *row = X.row[lane];
    uint32_t blk = 0;
    bool active = S.t < end;
    for (;;) {
        const double ahead = (end - S.t) / gap - 2.0;
        uint32_t safe4 = (active && ahead >= 4.0) ? (uint32_t)fmin(ahead, (double)round_packets) >> 2 : 0u;
        uint32_t budget4 = active ? round_packets / 4 - safe4 : 0u;
        uint32_t w[4];
        philox_c(blk, mi, episode, gid, key0, key1, w);
        // whole blocks without the exit test, wave-uniform trip count: a lane that has run out idles (and publishes nothing)
        while (__ballot(safe4 != 0u)) {
            uint32_t wn[4];
            const bool on = safe4 != 0u;
            philox_c(blk + (on ? 1u : 0u), mi, episode, gid, key0, key1, wn);
            const uint32_t a_old4 = S.a4;
            uint32_t jk = 0;
            if (on) {
#pragma unroll
                for (int k = 0; k < 4; k++) packet_t(S, always || w[k] < thr, dl, maxq, ebw, gap, base, dmask_b, cap_b, row, jk);
                blk++;
                safe4--;
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = wn[k];
            }
            flush_block(X, lane, base, a_old4, mask_b, jk);
        }
        while (__ballot(budget4 != 0u && S.t < end)) {
            const bool on = budget4 != 0u && S.t < end;
            const uint32_t a_old4 = S.a4;
            uint32_t jk = 0;
            if (on) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k > 0 && !(S.t < end)) break;
                    packet_t(S, always || w[k] < thr, dl, maxq, ebw, gap, base, dmask_b, cap_b, row, jk);
                }
                blk++;
                budget4--;
                philox_c(blk, mi, episode, gid, key0, key1, w);
            }
            flush_block(X, lane, base, a_old4, mask_b, jk);
        }
        active = active && S.t < end;
        if (!__ballot(active)) break;
    }
    L.q = S.q; L.tu = S.tu; L.t = S.t; L.a = S.a4 >> 4; L.d = S.d4 >> 4;
}


// ---- variant 8: write-behind.  A compute wavefront only WRITES its accepted records into an LDS ring (no read, no wait on the
// address path: dropped records still leave at once); a drain wavefront of the same workgroup reads the ring transposed -- lanes
// 4 e .. 4 e + 3 take the (up to) four records of env e's block -- and issues the global stores: 16 envs x <= 64 consecutive ring
// bytes per store instruction, and the stalls on the compute unit's address path are the drain wavefront's, not the packet chain's.
constexpr int kWbDepth = 4;   // blocks of four packets in flight between a compute wavefront and its drain wavefront
struct WbLds {
    double2 ring[kWbDepth][64][4];   // [block % depth][lane][j-th accepted record of the block]
    uint2 meta[kWbDepth][64];        // a_old4 (byte offset of the block's first accepted record before masking), count
    unsigned long long base[64];     // the lanes' ring bases
    uint32_t prod, cons, total;      // blocks published / drained; total blocks (0xFFFFFFFF until the compute wavefront is done)
    uint32_t pad;
};
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p) { return *(volatile const uint32_t *)p; }
__device__ __forceinline__ void packet_wb(PacketState &S, const bool lost, const double dl, const double maxq, const double ebw, const double gap,
                                          char *base, const uint32_t dmask_b, const uint32_t cap_b, double2 *row, double2 *trash, uint32_t &jk) {
    const double t = S.t;
    const double qcur = max0(S.q - (t - S.tu));
    const double grown = qcur + ebw;
    const double lat0 = dl + qcur;
    const double lim = __hiloint2double(lost ? (int)0xBFF00000u : __double2hiint(maxq), __double2loint(maxq));
    const bool dropped = grown > lim;
    const double keep = lost ? S.q : qcur;
    S.q = dropped ? keep : grown;
    S.tu = lost ? S.tu : t;
    double2 rec;
    rec.x = t + lat0;
    rec.y = lat0;
    *(dropped ? trash : row + jk) = rec;
    if (dropped) st_rec(reinterpret_cast<double2 *>(base + cap_b + (S.d4 & dmask_b)), rec);
    const uint32_t inc = dropped ? 16u : 0u;
    jk += dropped ? 0u : 1u;
    S.d4 += inc;
    S.a4 += 16u - inc;
    S.t = t + gap;
}
__device__ __forceinline__ void wb_publish(WbLds &X, const uint32_t lane, const uint32_t blkno, const uint32_t a_old4, const uint32_t jk) {
    X.meta[blkno % kWbDepth][lane] = make_uint2(a_old4, jk);
    asm volatile("" ::: "memory");           // (LDS accesses of a wavefront execute in order: the count below follows the records)
    if (lane == 0) *(volatile uint32_t *)&X.prod = blkno + 1u;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wb_wait_slot(WbLds &X, const uint32_t blkno) {   // the ring slot of block `blkno` is free again
    while (blkno >= lds_peek(&X.cons) + (uint32_t)kWbDepth) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void rounds_wb_compute(Lane &L, uint32_t key0, uint32_t key1, uint32_t round_packets, WbLds &X, const uint32_t lane) {
    const double dl = L.dl, maxq = L.maxq, ebw = L.ebw, gap = L.gap, end = L.end;
    const uint32_t thr = L.thr, gid = L.gid, episode = L.episode, mi = L.mi;
    const bool always = L.always;
    char *base = L.base;
    const uint32_t dmask_b = (2u * kCap - 1u) << 4, cap_b = kCap << 4;
    PacketState S;
    S.q = L.q; S.tu = L.tu; S.t = L.t; S.a4 = L.a << 4; S.d4 = L.d << 4;
    X.base[lane] = (unsigned long long)(uintptr_t)base;
    __shared__ double2 s_trash[8][64];
    double2 *trash = &s_trash[threadIdx.x >> 6][lane];
    uint32_t blk = 0, blkno = 0;
    bool active = S.t < end;
    for (;;) {
        const double ahead = (end - S.t) / gap - 2.0;
        uint32_t safe4 = (active && ahead >= 4.0) ? (uint32_t)fmin(ahead, (double)round_packets) >> 2 : 0u;
        uint32_t budget4 = active ? round_packets / 4 - safe4 : 0u;
        uint32_t w[4];
        philox_c(blk, mi, episode, gid, key0, key1, w);
        while (__ballot(safe4 != 0u)) {
            uint32_t wn[4];
            const bool on = safe4 != 0u;
            philox_c(blk + (on ? 1u : 0u), mi, episode, gid, key0, key1, wn);
            wb_wait_slot(X, blkno);
            double2 *row = X.ring[blkno % kWbDepth][lane];
            const uint32_t a_old4 = S.a4;
            uint32_t jk = 0;
            if (on) {
#pragma unroll
                for (int k = 0; k < 4; k++) packet_wb(S, always || w[k] < thr, dl, maxq, ebw, gap, base, dmask_b, cap_b, row, trash, jk);
                blk++;
                safe4--;
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = wn[k];
            }
            wb_publish(X, lane, blkno, a_old4, jk);
            blkno++;
        }
        while (__ballot(budget4 != 0u && S.t < end)) {
            const bool on = budget4 != 0u && S.t < end;
            wb_wait_slot(X, blkno);
            double2 *row = X.ring[blkno % kWbDepth][lane];
            const uint32_t a_old4 = S.a4;
            uint32_t jk = 0;
            if (on) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k > 0 && !(S.t < end)) break;
                    packet_wb(S, always || w[k] < thr, dl, maxq, ebw, gap, base, dmask_b, cap_b, row, trash, jk);
                }
                blk++;
                budget4--;
                philox_c(blk, mi, episode, gid, key0, key1, w);
            }
            wb_publish(X, lane, blkno, a_old4, jk);
            blkno++;
        }
        active = active && S.t < end;
        if (!__ballot(active)) break;
    }
    if (lane == 0) *(volatile uint32_t *)&X.total = blkno;
    L.q = S.q; L.tu = S.tu; L.t = S.t; L.a = S.a4 >> 4; L.d = S.d4 >> 4;
}
__device__ __forceinline__ void rounds_wb_drain(WbLds &X, const uint32_t lane) {
    const uint32_t mask_b = (kCap - 1u) << 4;
    const uint32_t r = lane & 3u;
    uint32_t b = 0;
    for (;;) {
        uint32_t have;
        for (;;) {
            have = lds_peek(&X.prod);
            if (have > b || lds_peek(&X.total) == b) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (have <= b) break;                 // drained everything the compute wavefront published, and it is done
        for (; b < have; b++) {
            const uint32_t slot = b % kWbDepth;
#pragma unroll
            for (uint32_t g = 0; g < 4u; g++) {
                const uint32_t sl = 16u * g + (lane >> 2);
                const uint2 M = X.meta[slot][sl];
                const double2 R = X.ring[slot][sl][r];
                const unsigned long long bs = X.base[sl];
                if (r < M.y) st_rec(reinterpret_cast<double2 *>(reinterpret_cast<char *>((uintptr_t)bs) + ((M.x + r * 16u) & mask_b)), R);
            }
            asm volatile("" ::: "memory");
            if (lane == 0) *(volatile uint32_t *)&X.cons = b + 1u;   // (the loads above have been issued: LDS executes them in order before a later write lands)
        }
    }
}
template <int DUMMY>
__global__ __launch_bounds__(512) void k_rounds_wb(const EnvP *P, EnvOut *O, char *rings, const uint32_t *perm, uint32_t key0, uint32_t key1, long long *ticks) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __shared__ WbLds s_wb[4];
    if (threadIdx.x < 4u) { s_wb[threadIdx.x].prod = 0; s_wb[threadIdx.x].cons = 0; s_wb[threadIdx.x].total = 0xFFFFFFFFu; }
    __syncthreads();
    if (wv >= 4u) { rounds_wb_drain(s_wb[wv - 4u], lane); return; }
    const uint32_t wave = blockIdx.x * 4u + wv;
    const uint32_t e = perm[wave * 64u + lane];
    Lane L;
    L.dl = P[e].dl; L.maxq = P[e].maxq; L.ebw = P[e].ebw; L.gap = P[e].gap; L.end = P[e].end; L.q = P[e].q; L.tu = P[e].tu; L.t = P[e].t;
    L.a = P[e].a; L.d = P[e].d; L.gid = P[e].gid; L.episode = P[e].episode; L.mi = P[e].mi;
    const double thr_d = ceil(P[e].lr * 4294967296.0);
    L.always = thr_d >= 4294967296.0;
    L.thr = L.always ? 0xFFFFFFFFu : (thr_d > 0.0 ? (uint32_t)thr_d : 0u);
    L.base = rings + (size_t)e * kRingBytes;
    const long long r0 = (long long)wall_clock64();
    rounds_wb_compute(L, key0, key1, 256u, s_wb[wv], lane);
    const long long r1 = (long long)wall_clock64();
    O[e].q = L.q; O[e].tu = L.tu; O[e].t = L.t; O[e].a = L.a; O[e].d = L.d;
    if (lane == 0) ticks[wave] = r1 - r0;
}

template <int V>
__global__ __launch_bounds__(256) void k_rounds(const EnvP *P, EnvOut *O, char *rings, const uint32_t *perm, int n_probe_waves, int neighbours,
                                                   uint32_t key0, uint32_t key1, long long *ticks) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + wv;
    const bool probe = wave < (uint32_t)n_probe_waves;
    if (!probe && (int)((wave - n_probe_waves) / n_probe_waves) >= neighbours) return;
    const uint32_t e = perm[wave * 64u + lane];
    Lane L;
    L.dl = P[e].dl; L.maxq = P[e].maxq; L.ebw = P[e].ebw; L.gap = P[e].gap; L.end = P[e].end; L.q = P[e].q; L.tu = P[e].tu; L.t = P[e].t;
    L.a = P[e].a; L.d = P[e].d; L.gid = P[e].gid; L.episode = P[e].episode; L.mi = P[e].mi;
    const double thr_d = ceil(P[e].lr * 4294967296.0);
    L.always = thr_d >= 4294967296.0;
    L.thr = L.always ? 0xFFFFFFFFu : (thr_d > 0.0 ? (uint32_t)thr_d : 0u);
    L.base = rings + (size_t)e * kRingBytes;
    __shared__ StageLds s_stage[4];
    const long long r0 = (long long)wall_clock64();
    if (!probe) rounds_v0<true>(L, key0, key1, 256u);
    else if (V == 0) rounds_v0<true>(L, key0, key1, 256u);
    else if (V == 6) rounds_v0<false>(L, key0, key1, 256u);
    else if (V == 7) rounds_t(L, key0, key1, 256u, s_stage[wv], lane);
    else rounds_bf<V>(L, key0, key1, 256u);
    const long long r1 = (long long)wall_clock64();
    O[e].q = L.q; O[e].tu = L.tu; O[e].t = L.t; O[e].a = L.a; O[e].d = L.d;
    if (lane == 0 && probe) ticks[wave] = r1 - r0;
}

}  // namespace

static double urand() { return (double)rand() / ((double)RAND_MAX + 1.0); }

int main(int argc, char **argv) {
    const int neighbours = argc > 1 ? atoi(argv[1]) : 0;
    const int n_probe = argc > 2 ? atoi(argv[2]) : 1024;   // 1 024 = one wavefront per SIMD; 256 = one per compute unit (workgroups of ONE wavefront then)
    const int wg_waves = n_probe >= 1024 ? 4 : 1;
    const int n_waves = n_probe * (1 + neighbours);
    const int n_env = n_waves * 64;
    std::vector<EnvP> hp(n_env);
    srand(12345);
    std::vector<int> packets(n_env);
    for (int i = 0; i < n_env; i++) {
        EnvP &p = hp[i];
        const double bw = 100.0 + 400.0 * urand();
        p.dl = 0.05 + 0.45 * urand();
        p.lr = urand() < 0.1 ? 0.0 : 0.05 * urand();
        p.maxq = (double)(1 + (int)exp(8.0 * urand())) / bw;   // queue in packets / bw
        p.ebw = 1.0 / bw;
        const double rate = bw * (0.3 + 1.5 * urand());
        p.gap = 1.0 / rate;
        const int n = 300 + rand() % 61;
        packets[i] = n;
        p.t = 3.0 + 10.0 * urand();
        p.end = p.t + ((double)n - 0.5) * p.gap;
        p.tu = p.t - p.gap;
        p.q = urand() < 0.5 ? 0.0 : p.maxq * urand();
        p.gid = (uint32_t)i; p.episode = 1; p.mi = 7; p.a = rand() % 1000; p.d = rand() % 1000;
    }
    std::vector<uint32_t> perm(n_env);
    for (int i = 0; i < n_env; i++) perm[i] = i;
    for (int i = n_env - 1; i > 0; i--) std::swap(perm[i], perm[rand() % (i + 1)]);
    const int mixed = argc > 3 ? atoi(argv[3]) : 0;   // arg 4 = 1: only wavefront 0 of a workgroup is long (300-360 packets), the others 40-80:
    if (mixed)                                        // the tail of the send launch, one long lane-round wavefront per compute unit
        for (int w = 0; w < n_waves; w++)
            for (int l = 0; l < 64; l++) {
                const int e = perm[w * 64 + l];
                const int n = (w % 4 == 0) ? packets[e] : 40 + rand() % 41;
                packets[e] = n;
                hp[e].end = hp[e].t + ((double)n - 0.5) * hp[e].gap;
            }
    const int lanes = argc > 4 ? atoi(argv[4]) : 64;  // arg 5: lanes of a wavefront that have an env with packets (the others' intervals are empty):
    if (lanes < 64)                                   // what would a lane-round item of 32 or 16 envs cost per iteration?
        for (int w = 0; w < n_waves; w++)
            for (int l = lanes; l < 64; l++) {
                const int e = perm[w * 64 + l];
                packets[e] = 0;
                hp[e].end = hp[e].t;
            }
    EnvP *dP; EnvOut *dO; char *rings; uint32_t *dperm; long long *dt;
    CK(hipMalloc(&dP, sizeof(EnvP) * n_env)); CK(hipMemcpy(dP, hp.data(), sizeof(EnvP) * n_env, hipMemcpyHostToDevice));
    CK(hipMalloc(&dO, sizeof(EnvOut) * n_env));
    CK(hipMalloc(&rings, kRingBytes * n_env));
    CK(hipMalloc(&dperm, 4 * n_env)); CK(hipMemcpy(dperm, perm.data(), 4 * n_env, hipMemcpyHostToDevice));
    CK(hipMalloc(&dt, 8 * n_probe));
    std::vector<EnvOut> ref;
    std::vector<char> ring_ref;
    const int probe_envs = n_probe * 64;
    for (int v : {0, 6, 1, 2, 5, 7, 8, 9}) {
        if (v == 8 && (wg_waves != 4 || neighbours)) continue;   // (write-behind: workgroups of 4 compute + 4 drain wavefronts)
        std::vector<long long> ht(n_probe);
        double best_med = 1e30, best_max = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(rings, 0, kRingBytes * n_env));
            CK(hipDeviceSynchronize());
#define L(V) hipLaunchKernelGGL(k_rounds<V>, dim3(n_waves / wg_waves), dim3(64 * wg_waves), 0, 0, dP, dO, rings, dperm, n_probe, neighbours, 0x1234u, 0x5678u, dt)
            if (v == 0) L(0); if (v == 1) L(1); if (v == 2) L(2); if (v == 3) L(3); if (v == 4) L(4); if (v == 5) L(5); if (v == 6) L(6); if (v == 7) L(7); if (v == 9) L(9);
            if (v == 8) hipLaunchKernelGGL(k_rounds_wb<0>, dim3(n_waves / 4), dim3(512), 0, 0, dP, dO, rings, dperm, 0x1234u, 0x5678u, dt);
#undef L
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ht.data(), dt, 8 * n_probe, hipMemcpyDeviceToHost));
            // ns per iteration of each wavefront: its longest lane's packets
            std::vector<double> per;
            for (int w = 0; w < n_probe; w++) {
                if (mixed && w % 4 != 0) continue;     // (the long wavefronts only)
                int mx = 0;
                for (int l = 0; l < 64; l++) mx = std::max(mx, packets[perm[w * 64 + l]]);
                per.push_back(ht[w] * 10.0 / mx);
            }
            std::sort(per.begin(), per.end());
            best_med = std::min(best_med, per[per.size() / 2]);
            best_max = std::min(best_max, per[per.size() - 1]);
        }
        // results: the probes' envs
        std::vector<EnvOut> ho(n_env);
        CK(hipMemcpy(ho.data(), dO, sizeof(EnvOut) * n_env, hipMemcpyDeviceToHost));
        std::vector<char> hr;
        size_t bad_state = 0, bad_ring = 0;
        if (v != 9 && v != 5 && v != 6) {
            // rings of the first 4 096 probe envs, byte for byte
            const int check = std::min(4096, probe_envs);
            hr.resize((size_t)check * kRingBytes);
            for (int j = 0; j < check; j++)
                CK(hipMemcpy(hr.data() + (size_t)j * kRingBytes, rings + (size_t)perm[j] * kRingBytes, kRingBytes, hipMemcpyDeviceToHost));
            if (v == 0) { ref = ho; ring_ref = hr; }
            for (int j = 0; j < probe_envs; j++) {
                const EnvOut &x = ho[perm[j]], &y = ref[perm[j]];
                if (memcmp(&x.q, &y.q, 8) || memcmp(&x.tu, &y.tu, 8) || memcmp(&x.t, &y.t, 8) || x.a != y.a || x.d != y.d) bad_state++;
            }
            bad_ring = memcmp(hr.data(), ring_ref.data(), hr.size()) ? 1 : 0;
        }
        printf("variant %d  probes %4d  neighbours %d  %6.1f ns per iteration (median wavefront)  %6.1f (slowest)  state mismatches %zu  rings %s\n", v, n_probe, neighbours,
               best_med, best_max, bad_state, (v == 9 || v == 5 || v == 6) ? "n/a" : (bad_ring ? "DIFFER" : "equal"));
    }
    return 0;
}

// sdust_sift.hpp — the round-3 decomposition of symmetric DUST for plain A/C/G/T sequence (included by sdust.hip, same
// anonymous namespace).  Replaces the per-lane recurrence of sdust_w64 wherever a chunk and the 128 bases in front of it
// hold nothing but A/C/G/T (either case); every other chunk is left to sdust_w64, which is exact for any input.
//
// What the reference computes there (src/sdust/sdust.c:104-128,:88-102, restated; tools/sim/sdust_sift_sim.c checks every
// statement below against the oracle for a dozen (T, W), chunk sizes down to W, mixed clean / unclean chunks):
//   * find_perfect over EVERY suffix of the window, without the gate of :149, is the same function, and the
//     result list is the canonical union (:94-98) of the intervals it inserts.
//   * An interval [s, i] (first base of its first word .. last base of its last word, l = words - 1, r = equal-word pairs)
//     is inserted iff it is PERFECT: 10 r > T l, and r / l >= the ratio of every sub-interval.  (Induction over the
//     sub-intervals: the running maximum of :113-118 is the best ratio among the inserted sub-intervals, and the best
//     sub-interval is itself inserted.)  So the result is a LOCAL function of the sequence: no state has to be carried.
//   * Necessary for [s, i] to be perfect (c_j = equal words in front of the j-th last word, inside the interval):
//       removing the last k words never raises the ratio:   sum_{j<k} 10 c_j > T k   for every k <= l,
//     and c_j <= ct(i - j), the equal words among the W - 3 words in front of position i - j.
// One kernel, one wave per chunk, three stages over the chunk and the two 64-base tiles in front of it (words in LDS):
//   sift     position-parallel: one lane per base of a tile.  ct(i) comes from two 64-entry tables of lane masks in LDS (this
//            tile, the tile before): every lane ORs its bit into the entry of its word, reads both entries and counts the bits
//            inside its window — five LDS operations per 64 bases instead of two counter updates per base.
//            Positions with ct > T / 10 (7 % of random sequence) are compacted, 64 at a time, into
//              L1: the partial sums above over the last min(lmin, 16) and 16 words (lmin = shortest candidate) -> 1.4 %
//              L2: an exact walk over the suffixes of up to 16 words (round 5: the 16 words packed in four registers, equal words
//                  counted by byte compares — no counters, no LDS atomics), or the 16-term sum for the longer ones -> 0.07 % of the
//                  positions of random sequence: one bit per base.  Any superset of the positions at which something is inserted
//                  would do.
//   resolve  the wave steps through the set bits with lane <-> age of the window's words (newest = lane 0), everything in
//            registers: word, suffix score r, P slot (ratio key | l << 24).  Consecutive positions are one incremental step
//            (equal-word ballot, mbcnt, three DPP shifts: ~11 vector instructions, ~30 with candidates); after a gap the window
//            is re-read from LDS and the slots shifted by the gap.  It starts W - 2 steps in front of its chunk with an EMPTY P:
//            an entry whose start is >= the first step - 2 only ever meets entries that were computed from there on, so it is
//            exact; entries are recorded by the time they leave the window (start + W), the rule the chunk rows and the stitch
//            already follow.
//   walk     A chunk that holds a byte other than A/C/G/T/a/c/g/t within [start - 128, end) is stepped through base by base by
//            the same wave with the same lane <-> age state, from the warm-up start sd_find_start() gives: the reference's loop
//            (:139-157) with its flush at a non-base and the window that lives on across it (P slots are keyed by start VALUE:
//            while a run is shorter than W behind a non-base the window start stands still and the slots do not age).
#pragma once

struct SiftArgs {
    const uint8_t *bases;
    const int64_t *ctg_off;
    const int32_t *ctg_len;
    const SdChunk *chunks;
    int32_t n_chunks;
    int32_t T, W;
    uint32_t lds_per_wave;    // bytes of dynamic LDS per wave
    uint32_t reg_cap;         // positions the word / count buffer holds (multiple of 64): chunk + 128
    // The chunks that hold other bytes than letters are stepped through base by base, the chunks inside repeat arrays position by
    // position: up to a millisecond of sequential work each.  The first call for a chunk table notes them (walk_out: [0] = count, [2 ..] ids; iswalk_out: a flag per chunk); from them the host makes `order`
    // (those chunks first, then the others), and later calls take chunk order[i] where they would have taken chunk i: the last
    // walk does not start when everything else is done.
    const uint32_t *order;
    uint32_t *walk_out;
    uint32_t *iswalk_out;
    uint32_t *counter;        // [SIFT_NCTR * 16], zero before the launch: counter c (at index 16 c) hands out the chunks c + j SIFT_NCTR
    int32_t thr, lmin;        // equal words a last word needs (T / 10 + 1); shortest l with 10 l (l + 1) / 2 > T l, capped at 16
    int32_t abl;              // development aid (CORNETTO_SIFT_ABL): 1 no resolve, 2 no L1 / L2, 4 no tiles: timing only, results are wrong
    int32_t dp_min;           // a tile with at least this many sifted positions is resolved end-parallel (dp tile); 65: never (CORNETTO_SIFT_DP)
    int32_t l2_skip;          // an L1 batch with at least this many survivors skips L2 (65: never; CORNETTO_SIFT_L2SKIP)
};

#ifndef SIFT_WPB
#define SIFT_WPB 1               // waves (= chunks) per workgroup; the waves share nothing
#endif
constexpr int SIFT_NCTR = 64;    // chunk counters, 64 bytes apart
constexpr int SIFT_PAD = 16;     // positions in front of the word / count buffer (look-back of L1 / L2 below offset 0)
#ifndef SIFT_K
#define SIFT_K 16                // L2: partial sums over this many counts, the exact walk over the suffixes of up to this many words
#endif
constexpr int SIFT_LS = SIFT_K - 1;   // L2 walks the suffixes with l <= SIFT_LS
// fixed part of the LDS of a wave: tables 4 x 64 x u32 | lists 2 x 128 x u16; then bits (cap / 8) and the word / count buffer, two
// bytes per position (word, count), and 16 bytes that the aligned dword reads of L2 may touch behind the last position.
// (Round 5: the 2 KB of per-lane nibble counters of the L2 walk are gone — the walk counts equal words among the 16 packed in four
// registers — and the columns of the dp tiles live in the two equal-word tables the resolve stage does not use: 7360 -> 5328 bytes
// for a 1536-base chunk = five 1280-byte granules instead of six, 25 waves per CU instead of 21.)
constexpr int SIFT_FIXED = 1024 + 512;
// (bits: the sifted positions, then the coverage of the region, cap / 8 bytes each)
__host__ __device__ constexpr uint32_t sift_lds_bytes(uint32_t cap) { return (SIFT_FIXED + cap / 4 + 2 * (SIFT_PAD + cap) + 16 + 15) / 16 * 16; }


// four bytes -> four codes: bits 0-1 the base (A0 C1 G2 T3), bit 2 set for anything that is not A/C/G/T/a/c/g/t
__device__ __forceinline__ uint32_t sd_codes4(uint32_t word)
{
    const uint32_t y = word & 0xDFDFDFDFu, idx = y & 0x07070707u;                    // fold case; low 3 bits: A1 C3 T4 G7
    const uint32_t code = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, idx);
    const uint32_t d = y ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, idx);     // the letter that must be there
    const uint32_t nz = ((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d;                        // bit 7 of a byte: byte != 0
    return code | ((nz >> 5) & 0x04040404u);
}
// the words that end at the four positions of `cn` (codes), `pcn` = the codes of the four positions before: bits 0-5 the
// word (:144), bit 6 = not a word (one of its three bases is not a base)
__device__ __forceinline__ uint32_t sd_words4(uint32_t cn, uint32_t pcn)
{
    const uint32_t y1 = __builtin_amdgcn_alignbyte(cn, pcn, 3);      // code of the previous position
    const uint32_t y2 = __builtin_amdgcn_alignbyte(cn, pcn, 2);      // and of the one before
    const uint32_t t = (cn & 0x03030303u) | ((y1 & 0x03030303u) << 2) | ((y2 & 0x03030303u) << 4);
    return t | (((cn | y1 | y2) & 0x04040404u) << 4);
}

__device__ __forceinline__ int sd_mbcnt64(unsigned long long m)       // set bits of m below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int sd_dpp_shr1(int old, int src, bool zero_fill)   // lane l <- lane l - 1; lane 0 <- 0 or `old`
{
    return zero_fill ? __builtin_amdgcn_update_dpp(0, src, 0x138, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false);
}

// ceil(2^32 / l) for l = 2 .. 63: floor(x * 2^13 / l) = umulhi(x << 13, this) for x < 2^11 (sd_ratio_key)
__device__ const uint32_t sd_recip_tab[64] = {0x0u, 0x0u, 0x80000000u, 0x55555556u, 0x40000000u, 0x33333334u, 0x2AAAAAABu, 0x24924925u, 0x20000000u, 0x1C71C71Du, 0x1999999Au, 0x1745D175u, 0x15555556u, 0x13B13B14u, 0x12492493u, 0x11111112u, 0x10000000u, 0xF0F0F10u, 0xE38E38Fu, 0xD79435Fu, 0xCCCCCCDu, 0xC30C30Du, 0xBA2E8BBu, 0xB21642Du, 0xAAAAAABu, 0xA3D70A4u, 0x9D89D8Au, 0x97B425Fu, 0x924924Au, 0x8D3DCB1u, 0x8888889u, 0x8421085u, 0x8000000u, 0x7C1F07Du, 0x7878788u, 0x7507508u, 0x71C71C8u, 0x6EB3E46u, 0x6BCA1B0u, 0x6906907u, 0x6666667u, 0x63E7064u, 0x6186187u, 0x5F417D1u, 0x5D1745Eu, 0x5B05B06u, 0x590B217u, 0x572620Bu, 0x5555556u, 0x539782Au, 0x51EB852u, 0x5050506u, 0x4EC4EC5u, 0x4D4873Fu, 0x4BDA130u, 0x4A7904Bu, 0x4924925u, 0x47DC120u, 0x469EE59u, 0x456C798u, 0x4444445u, 0x4325C54u, 0x4210843u, 0x4104105u};

// inclusive running maximum over the lanes, one v_max_u32 with a DPP source per step (lanes without a source in their row keep
// their value: bound_ctrl off); the s_nop are the two wait states a DPP read of a just-written register needs
__device__ __forceinline__ uint32_t sd_scan_max_dpp(uint32_t x)
{
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(x));
    return x;
}

// Where the base-by-base walk of a chunk starts: at or before the point sd_find_start() gives (W - 2 word emissions in front of
// chunk start - 2 W; starting earlier is exact as well).  That function is written for one lane among 64 busy ones: behind a run
// of non-bases it steps backwards one dependent byte load at a time — up to a millisecond when a whole wave waits for it.  Here
// the wave looks at 64 bases per load: a word ends at lane q >= 2 of a tile when lanes q - 2 .. q hold bases (words that begin in
// the tile before are not counted: the count can only be too small); the first tile border with W - 2 counted words behind it
// is the start.  Sequence without words for 4096 bases: the caller falls back to sd_find_start() and its word-count table.
__device__ __forceinline__ int sd_wave_find_start(const SdChunk ch, const uint8_t *seq, int W, int lane)
{
    if (ch.start <= 0) return 0;
    const int y = ch.start - 2 * W;
    if (y <= 2) return 0;
    int need = W - 2;
    for (int tile = (y - 1) >> 6, it = 0; tile >= 0 && it < 64; --tile, ++it) {
        const int p = tile * 64 + lane;
        const bool isb = p < y && nt4_code(seq[p]) < 4;
        const unsigned long long b = sd_ballot(isb);
        need -= __popcll(b & (b << 1) & (b << 2));
        if (need <= 0 || tile == 0) return tile * 64;
    }
    return -2;
}

// CAP: the positions the word / count buffer holds when they are known at compile time (SIFT_CAP_DEFAULT = the default chunk of SIFT_CHUNK_DEFAULT bases with the two
// tiles in front), else 0 = A.reg_cap.  With it every LDS address of a wave is the lane's part plus a literal: the build for a run-time size
// keeps dozens of derived addresses in scalar registers it does not have (84 spilled to vector-register lanes, a v_readlane per use:
// 5-8 of the 46 vector instructions per 64 bases, round 5).
// The default chunk: 1792 bases = 28 tiles, with the two in front 1920 positions — 5904 bytes of LDS per wave, the five 1280-byte granules that 1536 bases
// took as well (25 waves per CU), two staging rounds of 1024 bases.  Round 5, once the L2 counters were gone: per-chunk work (the flush of L1 / L2 with a third
// of a batch, staging, set-up: 19 of the 43 vector instructions per 64 bases) is spread over a sixth more bases — alone 4.53-4.69 against 4.62-4.84 ms
// (uniform), 11.9-12.2 against 12.2-12.5 (humanlike); in the step 6.47-6.59 against 6.66-6.71, 10.5 against 10.7, 14.2 against 14.5-15.0.  1920 bases (2048
// positions: still five granules) keeps a register in scratch; 2048 needs a sixth granule and a third staging round.
#ifndef SIFT_CHUNK_DEFAULT
#define SIFT_CHUNK_DEFAULT 1792
#endif
constexpr uint32_t SIFT_CAP_DEFAULT = SIFT_CHUNK_DEFAULT + 128;
template <bool STATS, uint32_t CAP = 0>
__global__ __launch_bounds__(64 * SIFT_WPB) __attribute__((amdgpu_waves_per_eu(CAP ? 7 : 6))) void sd_sift(SiftArgs A, SdArgs O)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sift_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = SIFT_WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // (uniform: the chunk table is read with scalar loads)
    uint8_t *const L = sift_lds + (size_t)wave * (CAP ? sift_lds_bytes(CAP) : A.lds_per_wave);
    // equal-word tables: [tile parity][half of the wave][word] -> the lanes of that half that hold the word.  32-bit entries, 64 per
    // table: the entry of word w lies in LDS bank w of every table.
    uint32_t *const tab = reinterpret_cast<uint32_t *>(L);
    uint16_t *const tl = reinterpret_cast<uint16_t *>(L + 1024);
    uint16_t *const tl2 = tl + 128;
    uint32_t *const sb = reinterpret_cast<uint32_t *>(L + SIFT_FIXED);
    const uint32_t cap = CAP ? CAP : A.reg_cap;
    // development aid (CORNETTO_SIFT_ABL: stages switched off for instruction counting, results are wrong): only in the build without a
    // compile-time buffer size — in the production build its seven tests were seven lane masks held in scalar registers
#ifdef CN_DEV
    const int abl = CAP ? 0 : A.abl;
#else
    constexpr int abl = 0;                            // (the ablations exist in the development build only: their results are wrong by design)
#endif
    // rounds of 1024 bases the staging of a region takes: with the default size the two rounds whose bases the chunk loop has fetched ahead
    // (no third round, no address for it held through the kernel)
    constexpr int STAGE_ROUNDS = CAP ? (int)((CAP + 1023) / 1024) : 4;
    uint32_t *const cb = sb + (cap >> 5);             // coverage of the region, one bit per base: what the chunk's rows are made of
    // the dp tiles keep the column of the position in front of a tile, per length, in the kilobyte of the equal-word tables: [length l] = c and
    // B of that position, ceil(2^32 / l), floor(T l / 10) — one 16-byte read per length gets all four.  The window reads of the stepping stage
    // (jump) use the first and the last quarter of the same kilobyte as their table pair and need them zero: they clear them when a dp tile
    // has been there (col_dirty).
    uint4 *const col = reinterpret_cast<uint4 *>(tab);
    uint8_t *const wc = L + SIFT_FIXED + cap / 4 + 2 * SIFT_PAD;
    const int T = A.T, W = A.W, CAPW = W - 2;

    // ---- per-lane constants -----------------------------------------------------------------------------------------
    // sift: which bits of the table entries lie inside the window of lane's position: positions i - (CAPW - 1) .. i - 1; in this
    // tile: lanes max(0, lane - (CAPW - 1)) .. lane - 1; in the tile before: the rest
    const int lo = lane - (CAPW - 1);
    const unsigned long long below = lane ? ~0ull >> (64 - lane) : 0ull;
    const unsigned long long mc = lo > 0 ? below & (~0ull << lo) : below;
    const unsigned long long mp = lo < 0 ? ~0ull << (64 + lo) : 0ull;
    // A lane of the lower half never needs the upper half of this tile's table (those lanes lie behind it), a lane of the upper half never
    // the lower half of the table before (W - 3 <= 63 positions back ends above lane 32 there): three gathers per tile, not four.  The four
    // 64-entry tables lie as [T0 lo][T1 lo][T1 hi][T0 hi] (T0 = the tiles of parity 0), which puts the "middle" table of a lane — the lower
    // half before / the upper half now — at one per-lane base plus a constant for either parity.
    const uint32_t m_a = (uint32_t)mc, m_d = (uint32_t)(mp >> 32), m_m = lane < 32 ? (uint32_t)mp : (uint32_t)(mc >> 32);
    const uint32_t mybit = 1u << (lane & 31);
    uint32_t *const mine0 = tab + (lane < 32 ? 0 : 192), *const mine1 = tab + (lane < 32 ? 64 : 128);   // tiles: this tile's table of this lane's half, parity 0 / 1
    uint32_t *const tab_mine = mine0;                         // resolve (window reads): the table of this lane's half, [T0 lo] / [T0 hi]
    const uint32_t *const mid1 = tab + (lane < 32 ? 0 : 128);   // the middle table at parity 1; at parity 0: + 64
    const int thr = A.thr, lmin = A.lmin;             // c > T / 10; the partial sums of L1 have lmin terms
    const int LS = CAPW - 1 < SIFT_LS ? CAPW - 1 : SIFT_LS;
    const bool long_ok = CAPW - 1 > LS;               // suffixes longer than the walk exist
    // statistics build: counted per wave, added up once at the end
    unsigned long long st_steps = 0, st_jumps = 0, st_cand = 0, st_trig = 0, st_l1 = 0, st_l2 = 0, st_walk = 0, st_tiles = 0, st_dp = 0, st_g2 = 0, st_g4 = 0, st_g8 = 0;

    struct Meta {
        SdChunk ch;
        int len;
        const uint8_t *seq;
    };
    struct Data {
        uint4 q[2];
        uint32_t pw[2];
    };
    auto meta = [&](int kk) {
        Meta m;
        m.ch = A.chunks[kk];
        m.len = A.ctg_len[m.ch.ctg];
        m.seq = A.bases + A.ctg_off[m.ch.ctg];
        return m;
    };
    auto fetch = [&](const Meta &m) {
        Data d;
        const int rb = m.ch.start - (m.ch.start > 0 ? 128 : 0), rlen = m.ch.end - rb;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int b16 = lane * 16 + it * 1024;
            d.q[it] = make_uint4(0, 0, 0, 0);
            if (b16 < rlen) d.q[it] = *reinterpret_cast<const uint4 *>(m.seq + rb + b16);
        }
        // The four bases in front of a lane's sixteen are the neighbouring lane's last dword (round 5: one DPP move instead of a second global
        // load per 16 bytes — a third of the kernel's vector-memory instructions); lane 0 takes lane 63's of the round before, and in the first
        // round the dword in front of the region, one scalar load (the region starts at a multiple of 64 inside the contig, or at its base 0).
        uint32_t first = 0;
        if (rb > 0) first = *reinterpret_cast<const uint32_t *>(m.seq + rb - 4);      // (uniform address)
        d.pw[0] = (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)d.q[0].w, 0x138, 0xF, 0xF, false);
        d.pw[1] = (uint32_t)__builtin_amdgcn_update_dpp(rdlane((int)d.q[0].w, 63), (int)d.q[1].w, 0x138, 0xF, 0xF, false);
        return d;
    };

  auto process = [&](const int k, const Meta &M, const Data &D) {
    const SdChunk ch = M.ch;
    const int len = M.len;
    const uint8_t *seq = M.seq;
    const int halo = ch.start > 0 ? 128 : 0;          // two tiles in front of the chunk: the first only feeds the "tile before" table
    const int rb = ch.start - halo;                   // region = [rb, ch.end), rb a multiple of 64
    const int rlen = ch.end - rb;
    const int ntile = (rlen + 63) >> 6;
    const bool islast = ch.end == len;
    const int rec_from = ch.start;

    // ---- stage the words of the region --------------------------------------------------------------------------------
    tab[lane] = 0;
    tab[64 + lane] = 0;
    tab[128 + lane] = 0;
    tab[192 + lane] = 0;
    for (int i = lane; i < (int)(cap >> 4); i += 64) sb[i] = 0;          // (the sifted positions and the coverage)
    if (lane < SIFT_PAD) reinterpret_cast<uint16_t *>(wc)[lane - SIFT_PAD] = halo ? 0xFF40 : 0x0040;   // no word; count unknown (large) / none
    bool bad = false;
    // Nearly every chunk holds letters only and lies inside its contig: the codes without the "not a base" flags, the words without the "no
    // word" bit, one accumulated difference to the letters that must be there — 12 instead of 25 vector instructions per dword.  Anything
    // else (another byte anywhere, the contig's first or last chunk) is staged again the general way below.
    bool general = rb <= 0 || rb + ((rlen + 15) & ~15) > len || (abl & 32);
    if (abl & 128) general = false;                 // (instruction counting only: no staging at all; with bit 4)
    if (!general && !(abl & 128)) {
        uint32_t acc = 0u;
        int lane_s = lane * 16;
        for (int b16 = lane * 16, it = 0; it < STAGE_ROUNDS && b16 < rlen; b16 += 1024, ++it) {
            uint4 q;
            uint32_t pw;
            if (STAGE_ROUNDS <= 2 || it < 2) {
                q = D.q[it];
                pw = D.pw[it];
            } else {
                // (regions beyond 2 KB — chunk sizes chosen by hand —: the address from a value the compiler cannot compute in front of the chunk
                // loop, where it would hold a vector register all through the tile loop for a path that the default chunk never takes)
                asm volatile("" : "+v"(lane_s));
                const uint8_t *const g = seq + rb + lane_s + it * 1024;
                q = *reinterpret_cast<const uint4 *>(g);
                pw = *reinterpret_cast<const uint32_t *>(g - 4);
            }
            const uint32_t ipw = pw & 0x07070707u;
            uint32_t pcn = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, ipw);
            acc |= (pw & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, ipw);
            const uint32_t in[4] = {q.x, q.y, q.z, q.w};
            uint32_t wo[8];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t idx = in[d] & 0x07070707u;
                const uint32_t cn = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, idx);
                acc |= (in[d] & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, idx);
                const uint32_t y1 = __builtin_amdgcn_alignbyte(cn, pcn, 3), y2 = __builtin_amdgcn_alignbyte(cn, pcn, 2);
                const uint32_t w4 = cn | y1 << 2 | y2 << 4;
                wo[2 * d] = __builtin_amdgcn_perm(0u, w4, 0x0C010C00u);
                wo[2 * d + 1] = __builtin_amdgcn_perm(0u, w4, 0x0C030C02u);
                pcn = cn;
            }
            uint4 *dst = reinterpret_cast<uint4 *>(wc + 2 * b16);
            dst[0] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
            dst[1] = make_uint4(wo[4], wo[5], wo[6], wo[7]);
        }
        general = sd_any(acc != 0u);
    }
    if (general)
    for (int b16 = lane * 16, it = 0; it < STAGE_ROUNDS && b16 < rlen; b16 += 1024, ++it) {
        const int p0 = rb + b16;
        uint4 q;
        uint32_t pw;
        if (STAGE_ROUNDS <= 2 || it < 2) {
            q = D.q[it];
            pw = D.pw[it];
        } else {                                      // (regions beyond 2 KB: chunk sizes chosen by hand)
            q = *reinterpret_cast<const uint4 *>(seq + p0);
            pw = *reinterpret_cast<const uint32_t *>(seq + p0 - 4);
        }
        uint32_t pcn = 0x04040404u;
        if (p0 > 0) pcn = sd_codes4(pw);
        const uint32_t in[4] = {q.x, q.y, q.z, q.w};
        uint32_t wo[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t cn = sd_codes4(in[d]);
            if (p0 + 4 * d + 4 > len) {               // (the contig's last dwords only)
                const int rem = len - (p0 + 4 * d);   // bytes of this dword inside the contig
                const uint32_t beyond = rem <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * rem);
                bad = bad || (cn & 0x04040404u & ~beyond) != 0u;
                cn |= 0x04040404u & beyond;
            } else {
                bad = bad || (cn & 0x04040404u) != 0u;
            }
            const uint32_t w4 = sd_words4(cn, pcn);
            wo[2 * d] = __builtin_amdgcn_perm(0u, w4, 0x0C010C00u);         // bytes 0, 1 -> halfwords
            wo[2 * d + 1] = __builtin_amdgcn_perm(0u, w4, 0x0C030C02u);     // bytes 2, 3
            pcn = cn;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(wc + 2 * b16);
        dst[0] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
        dst[1] = make_uint4(wo[4], wo[5], wo[6], wo[7]);
    }
    const bool unclean = sd_any(bad);
    SD_LDS_ORDER();

    // ---- state of the stepping stages (resolve, walk): lane <-> age of the window's words -----------------------------
    int w = 0, r = 0, slot = 0;                       // the word that entered `lane` steps ago, the score of the suffix that starts there, its P slot
    int sv0 = 0;                                      // start value (:121) of lane 0's slot: lane a holds the entry that starts at sv0 - a
    // the chunk's own list (:93-99); everything here is wave-uniform
    bool have_last = false;
    uint32_t last_s = 0, last_f = 0, n_out = 0;
    uint2 *const out = O.out + (size_t)k * O.cap;
    auto emit = [&](int ps, int pf, int tm) {
        if (tm < rec_from) return;
        if (have_last && ps <= (int)last_f) {
            if (pf > (int)last_f) last_f = (uint32_t)pf;
        } else {
            if (have_last) {
                if (n_out < O.cap && lane == 0) out[n_out] = make_uint2(last_s, last_f);
                ++n_out;
            }
            have_last = true;
            last_s = (uint32_t)ps;
            last_f = (uint32_t)pf;
        }
    };
    // every slot among lanes [from, CAPW), smallest start first, saved at time tm (:88-102 called with growing starts, :153)
    auto emit_slots = [&](int from, int tm) {
        unsigned long long em = sd_ballot(slot != 0 && lane >= from && lane < CAPW);
        while (em) {
            const int a = 63 - __builtin_clzll(em);
            em &= ~(1ull << a);
            const uint32_t so = (uint32_t)rdlane(slot, a);
            const int s = sv0 - a;
            emit(s, s + (int)(so >> 24) + 3, tm);
        }
    };
    // a word enters: the others age by one (those beyond amax leave); the slots age with them unless the window start stands still
    auto push_word = [&](int tw, int amax, bool age_slots) {
        const unsigned long long e = sd_ballot(w == tw) & ~(~1ull << (amax - 1 < 0 ? 0 : amax - 1)) & (amax > 0 ? ~0ull : 0ull);   // old ages 0 .. amax - 1 stay in the window
        r = sd_dpp_shr1(0, r, true) + sd_mbcnt64(e);                          // + equal words younger than the new age
        w = sd_dpp_shr1(tw, w, false);
        if (age_slots) slot = sd_dpp_shr1(0, slot, true);
    };
    // find_perfect (:104-128) over every suffix: lane = l; the running maximum of :113-118 is a scan of exact ratio keys
    auto pass = [&](int amax, const int Tl, const uint32_t m_recip) {
        // (lanes 1 .. amax: a scalar mask)
        const unsigned long long cm = sd_ballot(__mul24(r, 10) > Tl) & ~(~1ull << amax) & ~1ull;
        if (cm) {
            const bool cand = (cm >> lane) & 1ull;
            if (STATS) ++st_cand;
            uint32_t key_c = 0u;
            if (cand) key_c = lane == 1 ? ((uint32_t)r & 0x7FFu) << 13 : __umulhi(((uint32_t)r & 0x7FFu) << 13, m_recip);
            const uint32_t key_e = (uint32_t)slot & 0xFFFFFFu;
            const uint32_t xs = sd_scan_max_dpp(key_e > key_c ? key_e : key_c);
            const uint32_t sk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x138, 0xF, 0xF, true);
            const uint32_t km = sk > key_e ? sk : key_e;           // :113-117: entries with start >= this one
            if (cand && key_c >= km) slot = (int)(key_c | ((uint32_t)lane << 24));   // :118
        }
    };
    auto finish = [&]() {
        if (have_last) {
            if (n_out < O.cap && lane == 0) out[n_out] = make_uint2(last_s, last_f);
            ++n_out;
        }
        if (lane == 0) {
            O.out_n[k] = n_out;
            if (n_out > O.cap) atomicMax(O.ovf, n_out);
            if (STATS) st_tiles += (unsigned long long)ntile;
        }
    };

    if (unclean) {
        // stepping stages: lane = l (computed per chunk: two registers less across the tile loop)
        const int Tl = T * lane;
        const uint32_t m_recip = sd_recip_tab[lane];
        if (A.walk_out && lane == 0) {
            A.iswalk_out[k] = 1u;
            A.walk_out[2 + atomicAdd(&A.walk_out[0], 1u)] = (uint32_t)k;
        }
        // ---- walk: the reference's loop (:139-157), one base per step, from the warm-up start --------------------------
        int u = sd_wave_find_start(ch, seq, W, lane);
        if (u == -2) u = __builtin_amdgcn_readfirstlane(sd_find_start<true>(O, ch, seq));
        if (u < 0) return;                            // (the word-count table is needed: the host builds it and runs again)
        const int stop = islast ? len + 1 : ch.end;   // the contig's last chunk also takes the sentinel step i == len
        int l = 0, size = 0, cv = 4, ctile = -1;
        unsigned t = 0;
        for (int i = u; i < stop; ++i) {
            if ((i >> 6) != ctile) {
                ctile = i >> 6;
                const int p = ctile * 64 + lane;
                cv = p < len ? nt4_code(seq[p]) : 4;
            }
            const int b = rdlane(cv, i & 63);
            if (STATS) ++st_walk;
            if (b < 4) {
                ++l;
                t = (t << 2 | (unsigned)b) & 63u;                            // :144
                if (l >= 3) {
                    const int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);  // :146
                    const bool full = size >= CAPW;
                    // the window start moves on only in a run longer than W; then the oldest start leaves (:147).  While it stands
                    // still behind a non-base nothing leaves, and the entries keep their lanes: they are keyed by start VALUE
                    const bool moves = l > W;
                    if (full && moves) {
                        const uint32_t so = (uint32_t)rdlane(slot, CAPW - 1);
                        if (so) {
                            const int s = sv0 - (CAPW - 1);
                            emit(s, s + (int)(so >> 24) + 3, i);
                        }
                    }
                    size = full ? CAPW : size + 1;
                    push_word((int)t, size - 1, !full || moves);                // shift_window (:66-86)
                    sv0 = start + size - 1;
                    pass(size - 1, Tl, m_recip);
                }
            } else {
                emit_slots(0, i);                                               // :152-153
                slot = 0;
                l = 0;
                t = 0;                                                          // :154 — the window lives on
            }
        }
        finish();
        return;
    }

    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    int ntl = 0, ntl2 = 0;                            // (uniform) entries in the two lists
#ifdef SIFT_L2_ADAPT
    int l2_hot = 0, l2_skipped = 0;                   // (uniform) consecutive full L2 batches that kept nearly everything; L1 batches sifted without L2 since
#endif
    // L2: the 16-term sums (every candidate of more than 16 words needs them positive), the exact walk over the shorter suffixes.
    // Round 5: no counters.  The 16 (word, count) pairs that end at the lane's position are 32 consecutive bytes of the buffer: nine aligned
    // dword reads, realigned by the lane's 0 or 2 bytes and split into four registers of words and four of counts (oldest position in byte
    // 0 of register 0).  The suffix of a + 1 words has rr(a) = rr(a - 1) + [words among the a newer ones equal to the a-th last] equal-word
    // pairs: the a-th last word, replicated, is compared with the registers that hold newer words — (x ^ rep) + 0x3f per byte sets bit 6
    // exactly where the six-bit words differed — and the population count of those bits counts the unequal pairs: 36 register compares of
    // three vector instructions (v_xad, v_and, v_bcnt) for the fifteen steps, no LDS operation, no per-lane state.  (Before: sixteen returning LDS
    // atomics on 2 KB of 4-bit counters per wave, sixteen 16-bit reads and eight stores to clear them.)  A lane whose sixteen positions
    // hold one without a word — the first two of a contig, another byte in the halo — is kept as it is (any superset of the inserting
    // positions is exact), and so is every lane when the window has fewer than sixteen words.
    auto run_l2 = [&](int nb) {
        const bool on = lane < nb;
        const int o = on ? (int)tl2[lane] : 0;
        bool keep = on;
        if (LS == SIFT_LS) {
            const uint32_t at = (uint32_t)(uintptr_t)(lds_u32 *)wc + 2u * (uint32_t)o - 2u * (uint32_t)(SIFT_K - 1);   // LDS address of the oldest pair
            const lds_u32 *const pa = (const lds_u32 *)(uintptr_t)(at & ~3u);
            const uint32_t sh = at & 3u;                 // 0 or 2
            // The constants of the three-operand instructions (v_perm, v_bitop3: no literals on gfx950) are made HERE, in scalar registers: as
            // plain constants the compiler moves them into vector registers in front of the chunk loop — ten registers that the tile loop does
            // not have at six waves per SIMD (45 spilled registers, and the library refuses a build with scratch).
            uint32_t k80, ksel1, ksel2, ksel3, kw, kc;
            asm volatile("s_mov_b32 %0, 0x3f3f3f3f\n\ts_mov_b32 %1, 0x01010101\n\ts_mov_b32 %2, 0x02020202\n\ts_mov_b32 %3, 0x03030303\n\t"
                         "s_mov_b32 %4, 0x06040200\n\ts_mov_b32 %5, 0x07050301"
                         : "=s"(k80), "=s"(ksel1), "=s"(ksel2), "=s"(ksel3), "=s"(kw), "=s"(kc));
            uint32_t tt = (uint32_t)T;
            asm volatile("" : "+s"(tt));               // (the thresholds are made here, per batch: kept across the chunk loop they cost spilled scalar registers)
            // (the counts and their sums first, the words from a second read of the same dwords afterwards: four registers less at the peak)
            uint32_t Q[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t d0 = pa[2 * q], d1 = pa[2 * q + 1], d2 = pa[2 * q + 2];
                Q[q] = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(d2, d1, sh), __builtin_amdgcn_alignbyte(d1, d0, sh), kc);
            }
            // the sums: 10 S_k > k T  <=>  S_k > floor(k T / 10), k = 1 .. 16, newest count first
            // (every test is a compare into a lane mask that the scalar unit accumulates; the masks are tied to the vector values they were made
            // from, one step at a time — deferred, sixteen compares keep sixteen partial sums alive)
            uint32_t S = 0;
            unsigned long long pos_m = ~0ull;
#pragma unroll
            for (int a = 0; a < SIFT_K; ++a) {
                const int idx = SIFT_K - 1 - a;
                switch (idx & 3) {
                case 0: asm("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(S) : "v"(Q[idx >> 2])); break;
                case 1: asm("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(S) : "v"(Q[idx >> 2])); break;
                case 2: asm("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(S) : "v"(Q[idx >> 2])); break;
                default: asm("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(S) : "v"(Q[idx >> 2])); break;
                }
                const uint32_t th = (tt * (uint32_t)(a + 1)) / 10u;
                unsigned long long ok = __builtin_amdgcn_ballot_w64(S > th);
                asm volatile("" : "+s"(ok), "+v"(S));
                pos_m &= ok;
            }
            uint32_t at2 = at & ~3u;
            asm volatile("" : "+v"(at2), "+s"(pos_m));  // (the second read starts when the sums are through)
            const lds_u32 *const pb = (const lds_u32 *)(uintptr_t)at2;
            uint32_t P[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t d0 = pb[2 * q], d1 = pb[2 * q + 1], d2 = pb[2 * q + 2];
                P[q] = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(d2, d1, sh), __builtin_amdgcn_alignbyte(d1, d0, sh), kw);
            }
            const uint32_t seen = (P[0] | P[1] | P[2] | P[3]) & 0xC0C0C0C0u;
            // the walk: suffixes of 2 .. 16 words.  ne = the compared pairs that were NOT equal: rr = pairs - ne
            uint32_t ne = 0;
            unsigned long long sc_m = 0ull;
#pragma unroll
            for (int a = 1; a < SIFT_K; ++a) {
                const int idx = SIFT_K - 1 - a, qi = idx >> 2, bj = idx & 3;
                const uint32_t rep = __builtin_amdgcn_perm(P[qi], P[qi], bj == 0 ? 0u : bj == 1 ? ksel1 : bj == 2 ? ksel2 : ksel3);
#pragma unroll
                for (int q = qi; q < 4; ++q) {
                    if (q == qi && bj == 3) continue;                                    // (no newer word in its own register)
                    const uint32_t m = q == qi ? 0x40404040u << (8 * (bj + 1)) : 0x40404040u;   // the newer bytes of its own register / all of a newer one
                    const uint32_t u = (P[q] ^ rep) + k80;                              // (words are six bits: byte + 0x3f reaches bit 6 unless the two were equal, and never the next byte — one v_xad_u32)
                    ne += (uint32_t)__builtin_popcount(u & m);
                }
                // rr(a) = a (a + 1) / 2 - ne > floor(a T / 10)  <=>  ne < a (a + 1) / 2 - floor(a T / 10)   (never, when that is not positive)
                const int lim = a * (a + 1) / 2 - (int)((tt * (uint32_t)a) / 10u);
                unsigned long long hit = __builtin_amdgcn_ballot_w64((int)ne < lim);
                // (one step at a time — every step's inputs pass through here: scheduled freely, the compares of all fifteen steps start at once, twenty temporaries)
                asm volatile("" : "+s"(hit), "+v"(ne), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]));
                sc_m |= hit;
            }
            const bool sc = (sc_m >> lane) & 1ull, pos = (pos_m >> lane) & 1ull;
            keep = on && (seen != 0u || sc || (pos && long_ok));
        }
        if (STATS) st_l2 += (unsigned long long)__popcll(sd_ballot(keep));
#ifdef SIFT_L2_ADAPT
        if (nb == 64) {                               // inside a repeat the walk keeps what L1 kept: two such batches in a row switch it off for a while
            const int kept = __popcll(sd_ballot(keep));
            l2_hot = kept >= SIFT_L2_ADAPT ? (l2_hot < 2 ? l2_hot + 1 : 2) : 0;
            l2_skipped = 0;
        }
#endif
        if (keep) (void)__hip_atomic_fetch_or(&sb[o >> 5], 1u << (o & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // L1: the partial sums of 10 ct - T over the last k words must be positive for k = 1 .. lmin (every candidate has more words)
    auto run_l1 = [&](int nb) {
        const bool on = lane < nb;
        const int o = on ? (int)tl[lane] : 0;
        const uint8_t *const p = wc + 2 * o;
        int d = 0, dmin = 0x7fffffff;
        if (lmin == 4) {                              // (T = 16 .. 20, the default among them: the four reads in flight together)
            // 10 S_j > (j + 1) T  <=>  S_j > floor((j + 1) T / 10): an add and a compare per term, the floors by the scalar unit
            uint32_t tt = (uint32_t)T;
            asm volatile("" : "+s"(tt));
            const uint32_t c0 = p[1], c1 = p[-1], c2 = p[-3], c3 = p[-5];
            const uint32_t s1 = c0 + c1, s2 = s1 + c2, s3 = s2 + c3;
            dmin = ((c0 > tt / 10u) & (s1 > (2u * tt) / 10u) & (s2 > (3u * tt) / 10u) & (s3 > (4u * tt) / 10u)) ? 1 : 0;
        } else {
            for (int j = 0; j < lmin; ++j) {
                d += 10 * (int)p[1 - 2 * j] - T;
                dmin = d < dmin ? d : dmin;
            }
        }
        const bool ok = on && dmin > 0;
        const unsigned long long m = sd_ballot(ok);
#ifdef SIFT_L2_ADAPT
        const bool hot = l2_hot >= 2 && __popcll(m) >= 16;
        if (hot && ++l2_skipped >= 12) l2_hot = 1;    // (look again: the next full L2 batch decides)
        if (__popcll(m) >= A.l2_skip || hot) {
#else
        if (__popcll(m) >= A.l2_skip) {
#endif
            // most of a full batch passed: a repeat, where L2 keeps nearly everything as well and the dp tiles take any superset
            // at the same price — the survivors are sifted positions at once (any superset of the inserting positions is exact)
            if (STATS) st_l1 += (unsigned long long)__popcll(m);
            if (STATS) st_l2 += (unsigned long long)__popcll(m);
            if (ok) (void)__hip_atomic_fetch_or(&sb[o >> 5], 1u << (o & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (m) {
            if (STATS) st_l1 += (unsigned long long)__popcll(m);
            if (ok) tl2[ntl2 + sd_mbcnt64(m)] = (uint16_t)o;
            ntl2 += __popcll(m);
            SD_LDS_ORDER();
            if (ntl2 >= 64) {
                run_l2(64);
                const uint16_t mv = tl2[64 + lane];
                SD_LDS_ORDER();
                tl2[lane] = mv;
                ntl2 -= 64;
                SD_LDS_ORDER();
            }
        }
    };

    // ---- the tiles ---------------------------------------------------------------------------------------------------
    // PAR: parity of the tile (which pair of tables is "this tile").  GENERAL: lanes without a word (the first two positions of
    // a contig, behind its end) stay out of the tables.  TABLE_ONLY: the first tile in front of a chunk.
    uint8_t *pq = wc + 2 * lane;                      // this lane's position in the current tile
    // the three table addresses of a tile, one shift-add each (written out: the compiler shares the shift and pays an add per base, one of them
    // the addition of the block's LDS offset, which is zero)
    const uint32_t tab_a = (uint32_t)(uintptr_t)(lds_u32 *)tab, mine0_a = (uint32_t)(uintptr_t)(lds_u32 *)mine0, mine1_a = (uint32_t)(uintptr_t)(lds_u32 *)mine1,
                   mid1_a = (uint32_t)(uintptr_t)(lds_u32 *)mid1;
    auto lds_at = [](uint32_t wi, uint32_t base) __attribute__((always_inline)) {
        uint32_t a;
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(wi), "v"(base));
        return (lds_u32 *)(uintptr_t)a;
    };
    uint32_t wv_ahead = pq[0];
    auto tile = [&](auto par_c, auto general_c, auto only_c, const int q) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool GENERAL = decltype(general_c)::value, TABLE_ONLY = decltype(only_c)::value;
        constexpr int A_AT = PAR ? 64 : 0, D_AT = PAR ? 192 : 128, M_AT = PAR ? 0 : 64;     // this tile's lower half, the upper half before, the middle
        constexpr int Z0 = PAR ? 0 : 64, Z1 = PAR ? 192 : 128;                              // the two tables of the tile before
        // (the tile in front of a chunk as well: its first two words reach in front of the region, and a byte there that is not a
        // letter — it does not make the chunk unclean — leaves them without a word: bit 6)
        constexpr bool CHECK = GENERAL || TABLE_ONLY;
        const uint32_t wv = wv_ahead;                   // (read one tile ahead: the word bytes do not change while the tiles run, and the
        wv_ahead = pq[128];                           //  next tile's first LDS round trip overlaps this tile's; behind the region: nobody's bytes)
        asm("" : "+v"(wv_ahead));                      // (a 32-bit value as it comes from the byte load: no second zero extension)
        const bool valid = !CHECK || (wv < 64u && q * 64 + lane < rlen);
        const uint32_t wi = CHECK ? wv & 63u : wv;
        lds_u32 *const t_mine = lds_at(wi, PAR ? mine1_a : mine0_a), *const t_all = lds_at(wi, tab_a), *const t_mid = lds_at(wi, mid1_a);
        if (valid) (void)__hip_atomic_fetch_or(t_mine, mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        SD_LDS_ORDER();
        const uint32_t e_a = t_all[A_AT], e_m = t_mid[M_AT], e_d = t_all[D_AT];
        SD_LDS_ORDER();
        tab[Z0 + lane] = 0;                            // the tables of the tile before become the tables of the next tile
        tab[Z1 + lane] = 0;
        uint32_t ctu = (uint32_t)__popc(e_a & m_a);
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(ctu) : "v"(e_m & m_m));      // (the count's own adder: the compiler sums three counts with a fourth instruction)
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(ctu) : "v"(e_d & m_d));
        const int ct = (int)ctu;
        pq[1] = (uint8_t)(TABLE_ONLY ? 255 : (valid ? ct : 0));      // (no word there: nothing to count)
        pq += 128;
        if (!TABLE_ONLY) {
            const bool trig = valid && ct >= thr;
            const unsigned long long m = sd_ballot(trig);
            if (m) {
                if (STATS) st_trig += (unsigned long long)__popcll(m);
                // (every lane stores, the others into one entry behind the list: a select instead of an exec-mask region — three scalar
                // instructions of a loop that issues more scalar than vector instructions)
                int idx;                                   // trig ? its rank among the sifted lanes : 64 — entry ntl + 64 lies behind everything the list holds
                asm("v_cndmask_b32_e64 %0, 64, %1, %2" : "=v"(idx) : "v"(sd_mbcnt64(m)), "s"(m));
                (tl + ntl)[idx] = (uint16_t)(q * 64 + lane);
                ntl += __popcll(m);
                SD_LDS_ORDER();
                if (ntl >= 64) {
                    if (!(abl & 2)) run_l1(64);
                    const uint16_t mv = tl[64 + lane];
                    SD_LDS_ORDER();
                    tl[lane] = mv;
                    ntl -= 64;
                    SD_LDS_ORDER();
                }
            }
        }
    };
    if (!(abl & 4)) {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int q = 0;
        if (halo) tile(P0{}, std::false_type{}, std::true_type{}, q++);        // (a halo lies inside the contig: every lane has a word)
        else tile(P0{}, std::true_type{}, std::false_type{}, q++);
        // pairs of whole tiles, then what is left: at most one whole tile and the contig's last, partial one
        for (; q + 1 < ntile - 1; q += 2) {
            tile(P1{}, std::false_type{}, std::false_type{}, q);
            tile(P0{}, std::false_type{}, std::false_type{}, q + 1);
        }
        if (q < ntile - 1) { tile(P1{}, std::false_type{}, std::false_type{}, q); ++q; }
        if (q < ntile) {
            if (q & 1) tile(P1{}, std::true_type{}, std::false_type{}, q);
            else tile(P0{}, std::true_type{}, std::false_type{}, q);
        }
    }
    SD_LDS_ORDER();
    if (ntl > 0 && !(abl & 2)) run_l1(ntl);
    if (ntl2 > 0) run_l2(ntl2);
    if (abl & 1) { finish(); return; }
    SD_LDS_ORDER();

    // ---- resolve: the set bits in order ---------------------------------------------------------------------------------
    // What a clean chunk records: the inserted intervals whose start lies in [start - W, end - W) (the contig's first chunk: from 0, its
    // last: all) — the starts that leave the window at times inside the chunk, the rule of the walked chunks and of the stitch.  They
    // are marked in the coverage bits of the region when they are inserted (the result is the union of the inserted intervals: header);
    // the runs of the coverage are the chunk's rows.  No entry has to be followed until it leaves the window.
    const int s_lo = ch.start > 0 ? ch.start - W : -0x40000000, s_hi = islast ? 0x7fffffff : ch.end - W;
    // (made from a lane number the compiler cannot see through: as functions of `lane` alone the three would be computed once in front of the
    // chunk loop and occupy three vector registers all through the tile loop, which has none to spare)
    int lane_r = lane;
    asm volatile("" : "+v"(lane_r));
    const int Tl = T * lane_r;                        // stepping stages: lane = l
    const uint32_t m_recip = sd_recip_tab[lane_r];
    auto mark64 = [&](int o, int nbits) {             // bits [o, o + nbits) of the region, 1 <= nbits <= 64
        const uint32_t d = (uint32_t)o >> 5, b = (uint32_t)o & 31u;
        const unsigned long long mk = nbits >= 64 ? ~0ull : (1ull << nbits) - 1ull;
        const unsigned long long lo64 = mk << b;
        const uint32_t w1 = (uint32_t)(lo64 >> 32), w2 = b ? (uint32_t)(mk >> (64u - b)) : 0u;
        (void)__hip_atomic_fetch_or(&cb[d], (uint32_t)lo64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (w1) (void)__hip_atomic_fetch_or(&cb[d + 1], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (w2) (void)__hip_atomic_fetch_or(&cb[d + 2], w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto mark = [&](int s, int nbits) {               // the interval [s, s + nbits) of the contig, if this chunk records its start
        if (s < s_lo || s >= s_hi) return;
        if (nbits > 64) {                             // (W = 65, 66 only)
            mark64(s - rb, 64);
            mark64(s - rb + 64, nbits - 64);
        } else {
            mark64(s - rb, nbits);
        }
    };
    // find_perfect as `pass` above; an inserted interval is marked at once
    auto pass_m = [&](int amax) {
        const unsigned long long cm = sd_ballot(__mul24(r, 10) > Tl) & ~(~1ull << amax) & ~1ull;
        if (cm) {
            const bool cand = (cm >> lane) & 1ull;
            if (STATS) ++st_cand;
            uint32_t key_c = 0u;
            if (cand) key_c = lane == 1 ? ((uint32_t)r & 0x7FFu) << 13 : __umulhi(((uint32_t)r & 0x7FFu) << 13, m_recip);
            const uint32_t key_e = (uint32_t)slot;                 // (here a slot is the key alone: nothing follows an entry to its exit)
            const uint32_t xs = sd_scan_max_dpp(key_e > key_c ? key_e : key_c);
            const uint32_t sk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x138, 0xF, 0xF, true);
            const uint32_t km = sk > key_e ? sk : key_e;           // :113-117: entries with start >= this one
            if (cand && key_c >= km) {                              // :118
                slot = (int)key_c;
                mark(sv0 - lane, lane + 3);
            }
        }
    };

    unsigned long long sw = 0;
    if (lane < ntile) sw = (unsigned long long)sb[2 * lane] | (unsigned long long)sb[2 * lane + 1] << 32;
    // steps in front of start - W + 2 (of 2 at a contig start) are not taken: offsets below 128 - W + 2 (2)
    if (halo) {
        if (lane == 0) sw = 0;
        if (lane == 1) sw &= ~0ull << (66 - W);
    } else if (lane == 0) {
        sw &= ~3ull;
    }
    tab[lane] = 0;                                    // (the equal-word table of the window reads: the pair of parity 0)
    tab[192 + lane] = 0;
    SD_LDS_ORDER();

    int cur = 0;                                      // (a position of the contig)
    int n_dense = 0;                                  // steps taken in runs of consecutive positions
    bool have = false;                                // the slots are those of position cur
    bool fresh = false;                               // ... and so are w and r (not behind a dp tile)
    int dp_at = -1;                                   // the columns in LDS are those of this position
    int wt = 0, wt_tile = -1;                         // words of tile wt_tile of the region, lane <-> position
    // the state at position i, read from the staged words (a gap behind cur: the slots age by it); no pass
    bool col_dirty = false;                           // (uniform) the dp columns lie where jump's table pair must be zero
    auto jump = [&](int i) {
        if (col_dirty) {
            tab[lane] = 0;
            tab[192 + lane] = 0;
            col_dirty = false;
            SD_LDS_ORDER();
        }
        const int o = i - rb;
        const int amax = i - 2 < CAPW - 1 ? i - 2 : CAPW - 1;
        if (STATS) ++st_jumps;
        if (have) {
            const int g = i - cur;
            if (g > 0) {
                const int moved = __builtin_amdgcn_ds_bpermute(((lane - g) & 63) << 2, slot);
                slot = (g < CAPW && lane >= g) ? moved : 0;
            }
        } else {
            slot = 0;
        }
        // ---- the window at i; equal words at younger ages through the LDS table
        const bool ok = lane <= amax;
        w = ok ? (int)(wc[2 * (o - lane)] & 63u) : 0;      // (& 63: W = 65, 66 read the region's first two words at their first step, see above)
        if (ok) (void)__hip_atomic_fetch_or(&tab_mine[w], mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        SD_LDS_ORDER();
        const unsigned long long e = (unsigned long long)tab[w] | (unsigned long long)tab[192 + w] << 32;
        SD_LDS_ORDER();
        if (ok) tab_mine[w] = 0;
        SD_LDS_ORDER();
        r = wave_scan_add(ok ? sd_mbcnt64(e) : 0);
        cur = i;
        sv0 = i - 2;
        have = true;
        fresh = true;
    };
    // ---- dp tile: the 64 positions of tile t at once, lane <-> END of the interval, one step per length l (tools/sim/sdust_dp_sim.c):
    //   c(l, i) = equal words behind the first of the l + 1 words that end at i = c(l - 1, i - 1) + [word(i) == word(i - l)]
    //   r(l, i) = r(l - 1, i) + c(l, i);  key = the ratio r / l of a candidate (10 r > T l), 0 otherwise
    //   B(l, i) = max(key, B(l - 1, i), B(l - 1, i - 1)): the best candidate inside, itself included;  perfect <=> key >= the other two
    // Lane 0 takes its neighbour's values from the column of the position in front of the tile (ages there <-> lengths here);
    // lane 63 leaves the column of the tile's last position behind.  B starts at 1 (below every candidate's key), so that a key of
    // 0 never passes the test.  What a stepping stage behind the tile needs of P is the running maximum of :113-117 only — the best
    // entry among the starts of age <= a, for every a — and that is B of the last position: the slots are set to it.
    // Against one pass per position (~13 vector instructions per length and 64 positions instead of ~30 vector + ~30 scalar per
    // position) this pays from a dozen sifted positions per tile on.
    auto dp_tile = [&](int t) {
        const int o0 = 64 * t + lane;
        const uint8_t *const pw = wc + 2 * o0;
        const uint32_t w_own = pw[0];
        int c = 0, rr = 0;
        uint32_t B = 1u, lbest = 0u;
        const bool last = lane == 63;
        // one length: `cin` = the column entry of length l - 1 (lane 0's neighbour), `cl` = that of length l (its constants; lane 63 leaves c and B there).
        // (Round 5 tried the constants from lane l's registers — two v_readlane per length: 134 instead of 122 vector instructions per 64 bases on
        // the humanlike profile — and from scalar loads one turn ahead: 1 % slower still, A/B on one box.)
        auto step = [&](const int l, const uint4 cin, const uint4 cl) __attribute__((always_inline)) {
            const uint32_t wv = pw[-2 * l];
            const uint32_t rcp_l = cl.z;
            const int thr_l = (int)cl.w;
            c = __builtin_amdgcn_update_dpp((int)cin.x, c, 0x138, 0xF, 0xF, false) + (wv == w_own ? 1 : 0);
            rr += c;
            const uint32_t kq = __umulhi((uint32_t)rr << 13, rcp_l);
            const uint32_t key = rr > thr_l ? kq : 0u;
            const uint32_t Bn = (uint32_t)__builtin_amdgcn_update_dpp((int)cin.y, (int)B, 0x138, 0xF, 0xF, false);
            const uint32_t mx = B > Bn ? B : Bn;
            lbest = key >= mx ? (uint32_t)l : lbest;
            asm volatile("" : "+v"(lbest));           // (decided here: unrolled, the scheduler would keep key and mx of every length alive to the end)
            B = mx > key ? mx : key;
            SD_LDS_ORDER();
            if (last) *reinterpret_cast<uint2 *>(&col[l]) = make_uint2((uint32_t)c, B);
        };
        // (every column entry is read one step ahead of its use: lane 63 overwrites its c and B)
        uint4 ca = col[0], cb_ = col[1];
        {   // l = 1: the key is r itself (2^32 / 1 has no 32-bit reciprocal)
            const uint4 cn = col[2];
            const uint32_t wv = pw[-2];
            c = __builtin_amdgcn_update_dpp((int)ca.x, c, 0x138, 0xF, 0xF, false) + (wv == w_own ? 1 : 0);
            rr += c;
            const uint32_t key = rr > (int)cb_.w ? (uint32_t)rr << 13 : 0u;
            const uint32_t Bn = (uint32_t)__builtin_amdgcn_update_dpp((int)ca.y, (int)B, 0x138, 0xF, 0xF, false);
            const uint32_t mx = B > Bn ? B : Bn;
            lbest = key >= mx ? 1u : lbest;
            B = mx > key ? mx : key;
            SD_LDS_ORDER();
            if (last) *reinterpret_cast<uint2 *>(&col[1]) = make_uint2((uint32_t)c, B);
            ca = cb_;
            cb_ = cn;
        }
        // two lengths per turn: an entry is loaded into the register whose entry has just had its last use (no copies); ca = entry
        // l - 1, cb_ = entry l on entering
        int l = 2;
        for (; l + 1 < CAPW; l += 2) {
            step(l, ca, cb_);
            ca = col[l + 1];
            step(l + 1, cb_, ca);
            cb_ = col[l + 2 < 64 ? l + 2 : 63];
        }
        if (l < CAPW) step(l, ca, cb_);
        if (lbest) mark(rb + o0 - 2 - (int)lbest, (int)lbest + 3);
        SD_LDS_ORDER();
        slot = lane >= 1 && lane < CAPW ? (int)col[lane].y : 0;
        slot = slot == 1 ? 0 : slot;
        SD_LDS_ORDER();
    };
    const int dp_t1 = rlen >> 6;                       // whole tiles; the first two hold the steps with a short window
    unsigned long long nzt = sd_ballot(sw != 0ull);    // the tiles that hold a set bit at all (random sequence: 1 in 10)
    while (nzt) {
        const int t = __builtin_ctzll(nzt);
        nzt &= nzt - 1;
        unsigned long long m = rdlane64(sw, t);
        if (__popcll(m) >= A.dp_min && t >= 2 && t < dp_t1) {
            const int e = rb + 64 * t - 1;
            if (dp_at != e) {
                if (!(have && fresh && cur == e)) jump(e);
                // the columns at e: c(a) = r(a) - r(a - 1); B(a) = the best entry among the starts of age <= a
                // (the subtraction as written out: hipcc 7.2 folds `r - dpp(r)` into a v_sub_u32_dpp with its operands the wrong way round)
                const int rprev = sd_dpp_shr1(0, r, true);
                int cc;
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(cc) : "v"(r), "v"(rprev));
                const uint32_t bb = sd_scan_max_dpp((uint32_t)slot);
                // (with the constants of length `lane`: lane 63 rewrites c and B only)
                col[lane] = make_uint4((uint32_t)cc, bb > 1u ? bb : 1u, m_recip, (uint32_t)(Tl / 10));
                col_dirty = true;
                SD_LDS_ORDER();
            }
            if (lane == 0) *reinterpret_cast<uint2 *>(&col[0]) = make_uint2(0u, 1u);
            SD_LDS_ORDER();
            dp_tile(t);
            if (STATS) ++st_dp;
            n_dense += 64;
            cur = e + 64;
            dp_at = cur;
            have = true;
            fresh = false;
            continue;
        }
        while (m) {
            const int b = __builtin_ctzll(m);
            const int o = 64 * t + b;                 // offset in the region
            const int i = rb + o;
            if (have && fresh && i - cur == 1) {
                // ---- a run of consecutive positions: one step each — the words age by one, the new word comes in
                const unsigned long long inv = ~(m >> b);
                const int run = inv ? __builtin_ctzll(inv) : 64 - b;          // set bits from b on
                m &= run + b >= 64 ? ~(~0ull << b) : ~(~(~0ull << run) << b);
                if (t != wt_tile) {
                    wt_tile = t;
                    wt = wc[2 * (t * 64 + lane)];
                }
                n_dense += run;
                for (int j = 0; j < run; ++j) {
                    if (STATS) ++st_steps;
                    const int ii = i + j;
                    const int amax = ii - 2 < CAPW - 1 ? ii - 2 : CAPW - 1;
                    push_word(rdlane(wt, b + j), amax, true);
                    sv0 = ii - 2;
                    pass_m(amax);
                }
                cur = i + run - 1;
                continue;
            }
            m &= m - 1;
            if (STATS && have && fresh) { if (i - cur <= 2) ++st_g2; else if (i - cur <= 4) ++st_g4; else if (i - cur <= 8) ++st_g8; }
            jump(i);
            pass_m(i - 2 < CAPW - 1 ? i - 2 : CAPW - 1);
        }
    }
    // (a chunk inside a repeat array is a long sequential piece of work too: noted like the walked ones, handed out first next time)
    if (A.walk_out && n_dense >= 256 && lane == 0) {
        A.iswalk_out[k] = 1u;
        A.walk_out[2 + atomicAdd(&A.walk_out[0], 1u)] = (uint32_t)k;
    }
    // ---- the chunk's rows: the runs of the coverage (a run starts at a set bit behind a clear one, it ends in front of the next clear one)
    {
        SD_LDS_ORDER();
        const int nd = (rlen + 31) >> 5;
        uint32_t *const row = reinterpret_cast<uint32_t *>(out);
        uint32_t n_s = 0, n_e = 0;
        int carry = 0;                                // the dword in front of this round's first
        for (int base = 0; base <= nd; base += 64) {   // (one dword beyond the last: a run that reaches the region's end closes there)
            const int d = base + lane;
            const uint32_t v = d < nd ? cb[d] : 0u;
            const unsigned long long nz = sd_ballot(v != 0u);
            if (!nz && carry >= 0) {
                carry = 0;
                continue;
            }
            const uint32_t pm = (uint32_t)__builtin_amdgcn_update_dpp(carry, (int)v, 0x138, 0xF, 0xF, false) >> 31;
            const uint32_t sh = (v << 1) | pm;
            uint32_t st = v & ~sh, en = ~v & sh;
            const int ns = __popc(st), ne = __popc(en);
            const int xs = wave_scan_add(ns), xe = wave_scan_add(ne);
            uint32_t is = n_s + (uint32_t)(xs - ns), ie = n_e + (uint32_t)(xe - ne);
            while (st) {
                const int b = __builtin_ctz(st);
                st &= st - 1;
                if (is < O.cap) row[2 * is] = (uint32_t)(rb + 32 * d + b);
                ++is;
            }
            while (en) {
                const int b = __builtin_ctz(en);
                en &= en - 1;
                if (ie < O.cap) row[2 * ie + 1] = (uint32_t)(rb + 32 * d + b);
                ++ie;
            }
            n_s += (uint32_t)rdlane(xs, 63);
            n_e += (uint32_t)rdlane(xe, 63);
            carry = rdlane((int)v, 63);
        }
        if (lane == 0) {
            O.out_n[k] = n_s;
            if (n_s > O.cap) atomicMax(O.ovf, n_s);
            if (STATS) st_tiles += (unsigned long long)ntile;
        }
    }
  };

    // The waves stay (as many as the caller lets this kernel hold of the chip: the other stream needs its share) and take
    // chunks from SIFT_NCTR counters — counter c hands out the chunks c, c + SIFT_NCTR, c + 2 SIFT_NCTR, ... (a repeat array is
    // spread over all of them); a wave starts at counter (its number mod SIFT_NCTR) and moves on when one runs dry.  One counter
    // for everything took 14 ns per request: 1.8 M chunks = 25 ms.  The request for the next index is in flight while the
    // current chunk is worked on.
    int c = (int)((blockIdx.x * SIFT_WPB + wave) % SIFT_NCTR), dry = 0;
    auto ask = [&](int cc) {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&A.counter[cc * 16], 1u);
        return v;
    };
    uint32_t nxt = ask(c);
    for (;;) {
        const long long kk = (long long)__builtin_amdgcn_readfirstlane((int)nxt) * SIFT_NCTR + c;
        if (kk >= A.n_chunks) {
            // This counter is dry: a look at ALL of them at once, lane <-> counter (round 5; before: one request after the other at the next
            // counter until SIFT_NCTR of them had come back dry — at the end of the kernel every wave's 64 dependent atomic round trips, 7168 waves
            // on each address: 0.1 ms of a launch whatever its size, a seventh of the kernel of a 1/8 share).  Counters only grow, so one read at
            // or above its limit is final; one read below it is asked as before.
            static_assert(SIFT_NCTR == 64, "one lane per counter");
            if (dry < 2 * SIFT_NCTR) {
                const uint32_t seen = __hip_atomic_load(&A.counter[lane * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long left = (long long)A.n_chunks - lane;                    // counter `lane` hands out lane, lane + 64, ...: (left + 63) / 64 of them
                const unsigned long long open = sd_ballot(left > 0 && (long long)seen < (left + SIFT_NCTR - 1) / SIFT_NCTR);
                if (!open) break;
                ++dry;                                                                  // (a wave loses the race for a counter's last chunk once per counter at most)
                const int from = c + 1 == SIFT_NCTR ? 0 : c + 1;
                const unsigned long long rot = from ? (open >> from) | (open << (SIFT_NCTR - from)) : open;
                c = (from + __builtin_ctzll(rot)) & (SIFT_NCTR - 1);
                nxt = ask(c);
                continue;
            }
            // (not seen: reads that keep showing an open counter.  The exhaustive way: SIFT_NCTR dry answers in a row, one counter after the other)
            if (++dry >= 3 * SIFT_NCTR) break;
            c = c + 1 == SIFT_NCTR ? 0 : c + 1;
            nxt = ask(c);
            continue;
        }
        dry = 0;
        nxt = ask(c);
        const int ko = A.order ? __builtin_amdgcn_readfirstlane((int)A.order[kk]) : (int)kk;
        const Meta m = meta(ko);
        process(ko, m, fetch(m));
    }
    if (STATS && O.stats && lane == 0) {
        atomicAdd(&O.stats[0], st_steps);
        atomicAdd(&O.stats[1], st_jumps);
        atomicAdd(&O.stats[2], st_cand);
        atomicAdd(&O.stats[3], st_trig);
        atomicAdd(&O.stats[4], st_l1);
        atomicAdd(&O.stats[5], st_l2);
        atomicAdd(&O.stats[6], st_tiles);
        atomicAdd(&O.stats[7], st_walk);
        atomicAdd(&O.stats[8], st_dp);
        atomicAdd(&O.stats[9], st_g2);
        atomicAdd(&O.stats[10], st_g4);
        atomicAdd(&O.stats[11], st_g8);
    }
}

// pgq_walk.h — device helpers shared by the pair-centric kernels (pgq_meet.hip) and the neighbourhood-counting kernels
// (pgq_analytics.hip): a vertex set in LDS (open-addressing hash table + one-probe bit filter) and the segmented
// adjacency walk.
#pragma once
#include "pgq_internal.h"

namespace pgq {

typedef uint4 __attribute__((may_alias)) uint4_alias; // 16-byte clears of u32 tables (no type-based reordering)
constexpr int kMeetSlots = 1024;            // hash slots per wavefront
constexpr int kMeetSetMax = kMeetSlots / 2; // longest one-hop list the table takes (load factor <= 1/2)
constexpr u32 kMeetEmpty = 0xFFFFFFFFu;
constexpr int64_t kMeetOpen = -9;           // d_out marker: not answered here
constexpr int64_t kMeetOpen4 = -10;         // not answered, and distances 1..3 are excluded (k_meet3 finished its walk)

__device__ __forceinline__ u32 meet_hash(u32 x) { return (x * 0x9E3779B1u) >> 22; } // 10 bits

__device__ __forceinline__ bool meet_lookup(const u32 *tab, u32 x) {
	u32 h = meet_hash(x);
	for (;;) {
		const u32 t = tab[h];
		if (t == x) return true;
		if (t == kMeetEmpty) return false;
		h = (h + 1) & (kMeetSlots - 1);
	}
}

// one-probe pre-filter: 8192-bit map of the set (a set of ~100 vertices leaves ~1 % of the bits on), so that the four
// entries a lane holds cost four independent LDS reads instead of four dependent hash-table walks
constexpr int kMeetFilterWords = 256;
// the low 13 bits of the vertex id: one instruction; ids that differ by a multiple of 8192 share a bit, which costs a
// (rare) table walk, never a wrong answer
__device__ __forceinline__ u32 meet_fhash(u32 x) { return x & (kMeetFilterWords * 32 - 1); }
// `valid`: bit k set = entry k lies inside the segment.  Returns the entries that are in the set (bit k).  The four
// filter reads are independent and branch-free; only lanes holding a filter hit (~1 % of the entries) walk the table.
__device__ __forceinline__ u32 meet_probe4(const u32 *tab, const u32 *bm, const int4 v, u32 valid) {
	const u32 h0 = meet_fhash((u32)v.x), h1 = meet_fhash((u32)v.y), h2 = meet_fhash((u32)v.z), h3 = meet_fhash((u32)v.w);
	const u32 w0 = bm[h0 >> 5], w1 = bm[h1 >> 5], w2 = bm[h2 >> 5], w3 = bm[h3 >> 5];
	u32 p = (((w0 >> (h0 & 31)) & 1u) | (((w1 >> (h1 & 31)) & 1u) << 1) | (((w2 >> (h2 & 31)) & 1u) << 2) |
	         (((w3 >> (h3 & 31)) & 1u) << 3)) & valid;
	u32 f = 0;
	while (p) {
		const u32 k = (u32)__ffs((int)p) - 1u;
		p &= p - 1u;
		const u32 x = k == 0 ? (u32)v.x : (k == 1 ? (u32)v.y : (k == 2 ? (u32)v.z : (u32)v.w));
		if (meet_lookup(tab, x)) f |= 1u << k;
	}
	return f;
}

__device__ __forceinline__ u64 wave_min_u64(u64 x) {
	for (int o = 32; o > 0; o >>= 1) {
		const u64 y = __shfl_xor(x, o);
		x = y < x ? y : x;
	}
	return x;
}

// ---- packed walk over the padded adjacency (round 3) ---------------------------------------------------------------
// Device layout (pgq_runtime.hip, build_meet_layout): every vertex's list is copied into a padded adjacency whose lists
// start on a 16-byte group boundary (aligned to `meet_align` entries) and are filled up to whole groups with copies of
// their last entry, and every adjacency slot carries a 16-byte descriptor {neighbour, first group of the neighbour's
// padded list, its entries, 0}.  A two-hop walk therefore needs no offset look-ups (the expanded vertex's descriptor
// arrives with the one-hop list, streamed) and a lane's 16-byte load never straddles two lists, so ONE 64-lane request
// can carry groups of SEVERAL lists: the lists of a round of <= 64 expanded vertices form one virtual sequence of
// groups, lane l of request c reads virtual group 64 c + l; the lists drop their ids into a 64-byte LDS window at the
// positions where they begin and a DPP max-scan spreads them (seg_owner).  A 1-KB request then carries ~1 KB of list data whatever the list lengths (round 2: one request
// per list, ~35 % of the lanes useful on the SF100-shaped graph).  Measured alternatives that lost: a 6-step binary search per
// lane over the prefix sums with ds_bpermute (34 VALU + 7 dependent LDS round trips per request), a wave-uniform cursor
// over the list ends with one readlane + compare per boundary inside the request (k_meet3 0.267 vs 0.222 ms: a serial
// SALU/VALU chain with hazards), and 32-byte units per lane (one search per 2 KB, but 25 %
// more entries requested past the first hit and no less time).  Padding entries repeat a real entry of the same list:
// harmless for membership tests, marking and minima; callers that count must use `ok` and the per-entry length.
struct SegRound {
	u32 P;     // inclusive prefix sum of the round's group counts, per lane
	u32 D;     // first group of this lane's list minus the groups before it: virtual group x of the list sits at D + x
	u32 ng;    // groups of this lane's list
	u32 total; // groups in the round (wave-uniform)
};
// Inclusive scans over the 64 lanes with DPP (pure VALU: row shifts inside the rows of 16, then the two row
// broadcasts of gfx9): 7 instructions instead of 6 ds_bpermute round trips.  bound_ctrl reads 0 for lanes outside the
// row and lanes disabled by a row / bank mask keep `old` = 0, the identity of both operators (unsigned values).
#define PGQ_DPP(old, src, ctrl, rmask, bmask) (u32) __builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, rmask, bmask, true)
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 x) {
	u32 r = x + PGQ_DPP(0u, x, 0x111, 0xf, 0xf);  // row_shr:1
	r += PGQ_DPP(0u, x, 0x112, 0xf, 0xf);         // row_shr:2
	r += PGQ_DPP(0u, x, 0x113, 0xf, 0xf);         // row_shr:3
	r += PGQ_DPP(0u, r, 0x114, 0xf, 0xe);         // row_shr:4, lanes 4..15 of a row
	r += PGQ_DPP(0u, r, 0x118, 0xf, 0xc);         // row_shr:8, lanes 8..15 of a row
	r += PGQ_DPP(0u, r, 0x142, 0xa, 0xf);         // row_bcast:15 into rows 1 and 3
	r += PGQ_DPP(0u, r, 0x143, 0xc, 0xf);         // row_bcast:31 into rows 2 and 3
	return r;
}
__device__ __forceinline__ u32 wave_incl_max_u32(u32 x) {
	u32 r = max(x, PGQ_DPP(0u, x, 0x111, 0xf, 0xf));
	r = max(r, PGQ_DPP(0u, x, 0x112, 0xf, 0xf));
	r = max(r, PGQ_DPP(0u, x, 0x113, 0xf, 0xf));
	r = max(r, PGQ_DPP(0u, r, 0x114, 0xf, 0xe));
	r = max(r, PGQ_DPP(0u, r, 0x118, 0xf, 0xc));
	r = max(r, PGQ_DPP(0u, r, 0x142, 0xa, 0xf));
	r = max(r, PGQ_DPP(0u, r, 0x143, 0xc, 0xf));
	return r;
}
__device__ __forceinline__ SegRound seg_round(u32 gbeg, u32 len) {
	SegRound r;
	r.ng = (len + 3u) >> 2;
	r.P = wave_incl_scan_u32(r.ng);
	r.D = gbeg - (r.P - r.ng);
	r.total = (u32)__builtin_amdgcn_readlane((int)r.P, 63);
	return r;
}
// The list virtual group x0 + lane belongs to, for all 64 lanes of a request at once: every list that overlaps the
// window [x0, x0 + 64) drops its lane id (+1) at the window position where it begins (position 0 if it began earlier)
// into a 64-byte LDS window, and an inclusive max-scan over the positions spreads the ids to the right — two LDS
// operations and 7 DPP instructions instead of the 6 dependent ds_bpermute steps of a binary search per lane.  `win`:
// 64 zeroed bytes of this wavefront; they are zero again on return.  Lanes past the round's end get the last list.
__device__ __forceinline__ int seg_owner(const SegRound &r, u32 x0, unsigned char *win) {
	const int lane = threadIdx.x & 63;
	const u32 start = r.P - r.ng;
	if (r.ng != 0 && start < x0 + 64u && r.P > x0) win[start > x0 ? start - x0 : 0u] = (unsigned char)(lane + 1);
	__builtin_amdgcn_wave_barrier();
	const u32 own = win[lane];
	win[lane] = 0;
	return (int)wave_incl_max_u32(own) - 1;
}
typedef int pgq_v4i __attribute__((ext_vector_type(4)));
typedef unsigned int pgq_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 load_group_nt(const int32_t *__restrict__ xp, u32 group) {
	// streamed once: non-temporal, so that the lists do not push the descriptors of hot vertices out of L2
	const pgq_v4i r = __builtin_nontemporal_load(reinterpret_cast<const pgq_v4i *>(xp + (size_t)group * 4));
	return make_int4(r.x, r.y, r.z, r.w);
}

// Walks the padded lists of the descriptors list[0 .. list_n), DEPTH requests in flight, and calls f(v, ok, ev) per lane
// and request: v = one 16-byte group (four entries of ONE list), ok = the lane holds a group, ev = the expanded vertex
// the list belongs to (only when WANT_EV).  The descriptors are taken in rounds of 64 (lane j = descriptor j of the
// round); wavefront `w` of `nw` cooperating ones takes the requests w, w + nw, ... of every round, so the wavefronts of a
// workgroup share a round's groups evenly whatever the list lengths (a hub neighbour's list used to land on ONE of
// k_meet4d's 16 wavefronts: its phase took as long as that list).  `win`: 64 zeroed bytes of LDS owned by this wavefront
// (seg_owner).  stop() is wave-uniform and is asked after every DEPTH requests; `max_entries` bounds the entries
// requested (capped = true when it ended the walk).  `first` (have_first): the caller already holds descriptor `lane` of
// the first round (requested early, to overlap its latency).  Returns the entries of the requested groups (padding
// removed pro rata of the round; wave-uniform).
template <int DEPTH, bool WANT_EV, typename F, typename Stop>
__device__ __forceinline__ unsigned long long seg_walk(const uint4 *__restrict__ list, int list_n, int w, int nw,
                                                       const int32_t *__restrict__ xp, unsigned char *win, bool have_first,
                                                       uint4 first, unsigned long long max_entries, bool &capped, int &resume,
                                                       F f, Stop stop) {
	// Round 4: the DEPTH requests in flight are no longer tied to one round of 64 descriptors.  A request's slot keeps
	// everything needed to process it (its group, whether the lane holds one, the expanded vertex), so the slot freed by a
	// processed request is refilled from the NEXT round when the current one has nothing left for this wavefront: the
	// rounds overlap.  Before, every round ended with a drain — a dependent round trip per round, which is all a round
	// costs when its lists are short (R-MAT: a hub's 5000 neighbours are 78 rounds of mostly one request each = 78 round
	// trips; a mean-degree-89 list is two rounds).  The next round's descriptors are requested one round ahead as before.
	const int lane = threadIdx.x & 63;
	unsigned long long entries = 0, requested = 0; // requested: entries of the groups requested, against max_entries
	capped = false;
	resume = 0; // first descriptor of the earliest round with an unprocessed request when the walk ended
	const uint4 zero4 = make_uint4(0, 0, 0, 0);
	if (list_n <= 0) return 0;
	// ---- round state ----
	int pb = 0;
	uint4 d = zero4;
	if (have_first) d = first;
	else if (lane < list_n) d = list[lane];
	if (lane >= list_n) d = zero4;
	// the next round's descriptors: unconditional, the index clamped (a load under a per-lane condition is waited for at
	// the end of the branch)
	uint4 dn = zero4;
	if (64 < list_n) dn = list[min(64 + lane, list_n - 1)];
	SegRound r = seg_round(d.y, d.z);
	int nchunk = (int)((r.total + 63u) >> 6);
	int next = w;   // this wavefront's next request of the round
	int issued = 0; // ... and how many of the round's it has made
	bool open = true; // rounds are left
	// ---- requests in flight ----
	int4 x[DEPTH];
	int xc[DEPTH]; // < 0: empty; else the first descriptor of the request's round
	bool xok[DEPTH];
	u32 xv[DEPTH];
#pragma unroll
	for (int u = 0; u < DEPTH; u++) xc[u] = -1;
	auto issue = [&](int u) { // the current round's next request into slot u (caller: next < nchunk)
		const u32 xx = (u32)next * 64u + (u32)lane;
		const bool ok = xx < r.total;
		const u32 xs = ok ? xx : r.total - 1u; // lanes past the end re-read the last group (same line, masked by ok)
		const int j = seg_owner(r, (u32)next * 64u, win); // lanes past the end: the last list, like xs
		x[u] = load_group_nt(xp, (u32)__shfl((int)r.D, j) + xs);
		xok[u] = ok;
		if constexpr (WANT_EV) xv[u] = (u32)__shfl((int)d.x, j);
		else xv[u] = 0;
		xc[u] = pb;
		requested += 4ull * (unsigned long long)min(64u, r.total - (u32)next * 64u);
		next += nw;
		issued++;
	};
	bool halt = false;
	for (;;) {
		// top-up (and the first fill): slots left empty because the round had nothing more for this wavefront take
		// requests of the following rounds — the ONE place a round ends and the next begins
		while (open) {
			bool have_empty = false;
#pragma unroll
			for (int u = 0; u < DEPTH; u++) have_empty |= xc[u] < 0;
			if (!have_empty) break;
			if (next >= nchunk) {
				if (issued > 0) { // entries of the groups this wavefront requested in the round: pro rata of the round's groups
					u32 e = d.z;
					for (int o = 32; o > 0; o >>= 1) e += (u32)__shfl_xor((int)e, o);
					unsigned long long groups = (unsigned long long)issued * 64ull; // 64 per request, except the round's last one
					if (next - nw == nchunk - 1) groups -= (unsigned long long)nchunk * 64ull - r.total;
					entries += (unsigned long long)e * groups / r.total;
				}
				pb += 64;
				if (pb >= list_n) {
					open = false;
					break;
				}
				d = dn;
				if (pb + lane >= list_n) d = zero4;
				dn = zero4;
				if (pb + 64 < list_n) dn = list[min(pb + 64 + lane, list_n - 1)];
				r = seg_round(d.y, d.z);
				nchunk = (int)((r.total + 63u) >> 6);
				next = w;
				issued = 0;
				continue;
			}
#pragma unroll
			for (int u = 0; u < DEPTH; u++)
				if (xc[u] < 0 && next < nchunk) issue(u);
		}
		bool any_chunk = false;
#pragma unroll
		for (int u = 0; u < DEPTH; u++) {
			if (xc[u] < 0) continue; // wave-uniform
			any_chunk = true;
			const int4 v = x[u];
			const bool ok = xok[u];
			const u32 ev = xv[u];
			xc[u] = -1;
			if (open && next < nchunk) issue(u); // refilled before it is processed: everything about the request was copied above
			f(v, ok, ev);
		}
		if (!any_chunk) break;
		if (stop()) {
			halt = true;
			break;
		}
		if (requested > max_entries) {
			capped = true;
			halt = true;
			break;
		}
	}
	if (halt) {
		if (open && issued > 0) { // the round the walk stopped in
			u32 e = d.z;
			for (int o = 32; o > 0; o >>= 1) e += (u32)__shfl_xor((int)e, o);
			unsigned long long groups = (unsigned long long)issued * 64ull;
			if (next - nw == nchunk - 1) groups -= (unsigned long long)nchunk * 64ull - r.total;
			entries += (unsigned long long)e * groups / r.total;
		}
		resume = pb;
#pragma unroll
		for (int u = 0; u < DEPTH; u++)
			if (xc[u] >= 0) resume = min(resume, xc[u]);
	}
	return entries;
}

// ---- round 4: the set side as a two-bit filter in LDS + the ids themselves in registers ---------------------------------
// Round 3 kept the set (the non-expanded endpoint's one-hop list, <= 512 ids) in an open-addressing table of 4 KB next to
// a 2-KB filter: 6.2 KB of LDS per wavefront = 6.25 wavefronts per SIMD, and tools/membench's segment gather (1-KB
// segments out of a table far larger than the Infinity Cache) moves 3.7 TB/s at 4 wavefronts per SIMD and 5.9 TB/s at 8.
// The exact test does not need LDS: the ids sit in 8 registers per lane (lane l holds ids l, l + 64, ...), and a
// candidate the filter lets through — wave-uniform after a readlane — is compared with all 512 of them in 8 v_cmp.  The
// filter takes the whole 4 KB (32768 bits, both bits of an id in ONE word: word = id bits 5..14, bits = id bits 0..4 and
// 15..19, higher id bits folded in when V needs them), so a wavefront needs 4.2 KB and 8 fit a SIMD.  For V < 2^20 the
// filter is exact up to cross-combinations of two set members that share a word (~0.002 % of the probes for a 100-id set).
constexpr int kSetRegs = 8;                  // 64 x 8 = 512 ids in registers
constexpr int kSetRegMax = 64 * kSetRegs;
constexpr int kFltWords = 1024;              // 4 KB per wavefront
template <bool BIGV> __device__ __forceinline__ u32 flt_b(u32 x) { // only the low 5 bits are used (as a shift amount)
	return BIGV ? ((x >> 15) ^ (x >> 20) ^ (x >> 25)) : (x >> 15);
}
__device__ __forceinline__ u32 flt_word(u32 x) { return (x >> 5) & (kFltWords - 1); }
template <bool BIGV> __device__ __forceinline__ u32 flt_mask(u32 x) { return (1u << (x & 31)) | (1u << (flt_b<BIGV>(x) & 31)); }
// bit 0 of the result: both bits of x are set in its word w
template <bool BIGV> __device__ __forceinline__ u32 flt_test(u32 w, u32 x) { return (w >> (x & 31)) & (w >> (flt_b<BIGV>(x) & 31)); }
struct RegSet {
	u32 r[kSetRegs]; // lane l: ids l, l + 64, ... of the set list; kMeetEmpty past its end
	int rounds;      // registers in use (wave-uniform)
};
// exact membership of a wave-uniform id
__device__ __forceinline__ bool regset_has(const RegSet &s, u32 x) {
	bool in = false;
#pragma unroll
	for (int k = 0; k < kSetRegs; k++)
		if (k < s.rounds) in |= s.r[k] == x;
	return __any(in) != 0;
}
// the four filter words of a lane's group: independent LDS reads, no branches.  Bit k of the result: entry k passed.
template <bool BIGV> __device__ __forceinline__ u32 flt_pass4(const u32 *bm, const int4 v) {
	const u32 w0 = bm[flt_word((u32)v.x)], w1 = bm[flt_word((u32)v.y)], w2 = bm[flt_word((u32)v.z)], w3 = bm[flt_word((u32)v.w)];
	const u32 t0 = flt_test<BIGV>(w0, (u32)v.x), t1 = flt_test<BIGV>(w1, (u32)v.y);
	const u32 t2 = flt_test<BIGV>(w2, (u32)v.z), t3 = flt_test<BIGV>(w3, (u32)v.w);
	return (t0 & 1u) | ((t1 & 1u) << 1) | ((t2 & 1u) << 2) | ((t3 & 1u) << 3);
}
// Calls hit(x, L) for every entry the filter let through (p: bit k = entry k of this lane's group v) that IS in the set;
// x and the lane L holding it are wave-uniform.  The loop runs over the lanes with candidates only (none, nearly always) and
// ends when hit() returns true (the caller needs no further witness: on R-MAT-22 a request of 256 two-hop entries holds a
// hundred members of the set — the same few hubs — and verifying them all, one after the other, took 20-60 us per row).
template <typename H> __device__ __forceinline__ void verify_candidates(const RegSet &s, u32 p, const int4 &v, H hit) {
	u64 m = __ballot(p != 0);
	while (m) {
		const int L = __ffsll((long long)m) - 1;
		m &= m - 1;
		const u32 pl = (u32)__builtin_amdgcn_readlane((int)p, L);
		if (pl & 1u) {
			const u32 x = (u32)__builtin_amdgcn_readlane(v.x, L);
			if (regset_has(s, x) && hit(x, L)) return;
		}
		if (pl & 2u) {
			const u32 x = (u32)__builtin_amdgcn_readlane(v.y, L);
			if (regset_has(s, x) && hit(x, L)) return;
		}
		if (pl & 4u) {
			const u32 x = (u32)__builtin_amdgcn_readlane(v.z, L);
			if (regset_has(s, x) && hit(x, L)) return;
		}
		if (pl & 8u) {
			const u32 x = (u32)__builtin_amdgcn_readlane(v.w, L);
			if (regset_has(s, x) && hit(x, L)) return;
		}
	}
}

// Streams the adjacency segments of list[w], list[w + stride], ... (positions < list_n) and calls f(entry) for every
// entry until stop() (wave-uniform) says so.  Returns the number of entries requested.
template <typename F, typename Stop>
__device__ __forceinline__ unsigned long long meet_walk(const int32_t *__restrict__ list, int list_n, int w, int stride,
                                                        const int64_t *__restrict__ xoff, const int32_t *__restrict__ xadj,
                                                        F f, Stop stop) {
	const int lane = threadIdx.x & 63;
	unsigned long long entries = 0;
	for (int pb = w; pb < list_n; pb += 64 * stride) {
		int vb = 0, ve = 0;
		u32 vid = 0;
		const int p = pb + lane * stride;
		if (p < list_n) {
			vid = (u32)list[p];
			vb = (int)xoff[vid];
			ve = (int)xoff[vid + 1];
		}
		const int cnt = min(64, (list_n - pb + stride - 1) / stride);
		int j = -1, q = 0, e = 0, b = 0;
		u32 cv = 0;
		auto seek = [&]() {
			for (j++; j < cnt; j++) {
				b = __builtin_amdgcn_readlane(vb, j);
				e = __builtin_amdgcn_readlane(ve, j);
				if (e > b) {
					q = b & ~3;
					cv = (u32)__builtin_amdgcn_readlane((int)vid, j);
					return;
				}
			}
		};
		seek();
		constexpr int DEPTH = 4;
		int4 x[DEPTH];
		int xb[DEPTH], xe[DEPTH], xq[DEPTH];
		u32 xv[DEPTH];
		auto fetch = [&](int u) {
			xq[u] = -1;
			if (j < cnt) {
				const int t = q + 4 * lane;
				// unconditional (a load under a per-lane condition is waited for at the end of the branch): lanes past the
				// segment re-read its first group, masked by the range tests below
				{ // non-temporal: the lists are streamed once, the offset look-ups should stay in L2
					typedef int v4i __attribute__((ext_vector_type(4)));
					const v4i r = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(xadj + (t < e ? t : q)));
					x[u] = make_int4(r.x, r.y, r.z, r.w);
				}
				xb[u] = b;
				xe[u] = e;
				xq[u] = q;
				xv[u] = cv;
				entries += (unsigned long long)(min(e, q + 256) - max(b, q));
				q += 256;
				if (q >= e) seek();
			}
		};
#pragma unroll
		for (int u = 0; u < DEPTH; u++) fetch(u);
		for (;;) {
			bool any_chunk = false;
#pragma unroll
			for (int u = 0; u < DEPTH; u++) {
				if (xq[u] < 0) continue;
				any_chunk = true;
				const int4 v = x[u];
				const int t = xq[u] + 4 * lane, sb = xb[u], se = xe[u];
				const u32 ev = xv[u];
				fetch(u);
				if (t >= sb && t < se) f((u32)v.x, ev);
				if (t + 1 >= sb && t + 1 < se) f((u32)v.y, ev);
				if (t + 2 >= sb && t + 2 < se) f((u32)v.z, ev);
				if (t + 3 >= sb && t + 3 < se) f((u32)v.w, ev);
			}
			if (!any_chunk || stop()) break;
		}
		if (stop()) break;
	}
	return entries;
}

// The same walk with a per-chunk callback: g(v, valid, ev) gets the four entries a lane holds (bit k of `valid` set =
// entry k lies inside its segment) so that it can issue its own memory operations four at a time.
template <int DEPTH, typename G, typename Stop>
__device__ __forceinline__ unsigned long long meet_walk_chunks(const int32_t *__restrict__ list, int list_n, int w, int stride,
                                                               const int64_t *__restrict__ xoff,
                                                               const int32_t *__restrict__ xadj, G g, Stop stop) {
	const int lane = threadIdx.x & 63;
	unsigned long long entries = 0;
	for (int pb = w; pb < list_n; pb += 64 * stride) {
		int vb = 0, ve = 0;
		u32 vid = 0;
		const int p = pb + lane * stride;
		if (p < list_n) {
			vid = (u32)list[p];
			vb = (int)xoff[vid];
			ve = (int)xoff[vid + 1];
		}
		const int cnt = min(64, (list_n - pb + stride - 1) / stride);
		int j = -1, q = 0, e = 0, b = 0;
		u32 cv = 0;
		auto seek = [&]() {
			for (j++; j < cnt; j++) {
				b = __builtin_amdgcn_readlane(vb, j);
				e = __builtin_amdgcn_readlane(ve, j);
				if (e > b) {
					q = b & ~3;
					cv = (u32)__builtin_amdgcn_readlane((int)vid, j);
					return;
				}
			}
		};
		seek();
		int4 x[DEPTH];
		int xb[DEPTH], xe[DEPTH], xq[DEPTH];
		u32 xv[DEPTH];
		auto fetch = [&](int u) {
			xq[u] = -1;
			if (j < cnt) {
				const int t = q + 4 * lane;
				typedef int v4i __attribute__((ext_vector_type(4)));
				const v4i r = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(xadj + (t < e ? t : q)));
				x[u] = make_int4(r.x, r.y, r.z, r.w);
				xb[u] = b;
				xe[u] = e;
				xq[u] = q;
				xv[u] = cv;
				entries += (unsigned long long)(min(e, q + 256) - max(b, q));
				q += 256;
				if (q >= e) seek();
			}
		};
#pragma unroll
		for (int u = 0; u < DEPTH; u++) fetch(u);
		for (;;) {
			bool any_chunk = false;
#pragma unroll
			for (int u = 0; u < DEPTH; u++) {
				if (xq[u] < 0) continue;
				any_chunk = true;
				const int4 v = x[u];
				const int t = xq[u] + 4 * lane, sb = xb[u], se = xe[u];
				const u32 ev = xv[u];
				fetch(u);
				const u32 valid = (u32)(t >= sb && t < se) | ((u32)(t + 1 >= sb && t + 1 < se) << 1) |
				                  ((u32)(t + 2 >= sb && t + 2 < se) << 2) | ((u32)(t + 3 >= sb && t + 3 < se) << 3);
				g(v, valid, ev);
			}
			if (!any_chunk || stop()) break;
		}
		if (stop()) break;
	}
	return entries;
}

} // namespace pgq

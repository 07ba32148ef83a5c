// pgq_ball.h — source-centric search for rows that arrive GROUPED BY SOURCE (round 6; included by pgq_meet.hip, whose
// launch chain it opens).
//
// What it answers is what IterativeLengthFunction reports (iterativelength.cpp:34-143): the BFS distance of (src, dst), a
// pure function of (CSR, src, dst).  The binder evaluates `iterativelength` on a CROSS PRODUCT of endpoints
// (match.cpp:467-495, :658-671): a nested-loop join emits every source's rows in one stretch.  The reference spends one
// lane per source and sweeps the whole graph per level (iterativelength.cpp:12-32); rounds 1-5 did the same on the GPU
// (2048-lane batches: 40 M in-edges x a 256-byte frontier row per dense level).  But all rows of one source share ONE
// forward ball, and on a small-world graph a radius-2 ball is a few thousand vertices:
//
//     S1 = N_out(s), S2 = S1 + N_out(S1)           one vertex bit map per workgroup, in LDS (global slice when V is large)
//     d == s -> 0;  d in S1 -> 1;  d in S2 -> 2
//     else some in-neighbour of d in S2 -> 3       (d is not within 2, so that in-neighbour is at distance exactly 2)
//     else some in-neighbour of an in-neighbour of d in S2 -> 4
//     else: left open (distance >= 5, unreachable, or over a cap) -> the older routes
//
// One 1024-thread workgroup per SEGMENT = a stretch of at most 1024 consecutive rows with one source inside an aligned
// 1024-row window (k_ball_segments lists them): thread t owns row t of the segment; the two-hop ball is marked by the 16
// wavefronts together (seg_walk over the source's slot descriptors, pgq_walk.h), the in-lists of the destinations are
// scanned 16 lanes per row (64 entries per step, four rows per wavefront and step), the few rows at distance 4 get a
// wavefront each.  Per segment ~4 B x (two-hop walk of the source) + per row ~4 B x (in-degree of the destination): the
// SF100-shaped knows graph, 2048 sources x 1024 destinations: ~0.9 GB instead of the 9.4 GB of the lane batches.
// Whether the call takes this route is decided ON THE DEVICE from the exact number of source runs (k_ball_segments' last
// workgroup): bytes of the balls + row scans against the pre-pass's bytes per row and the lane batches' level bytes.
#pragma once

namespace pgq {

constexpr int kBallRows = 1024;  // rows of a segment = threads of a k_src_ball workgroup
constexpr int kBallOpenI = -9;   // per-row state in LDS: not answered (yet)

struct BallRule {
	double seg_floor;   // the least a segment costs, in bytes at streaming rate: its dependent round trips on one of the chip's workgroup slots
	double seg_bytes;   // bytes one ball is priced at: 4 x the mean two-hop walk of a vertex + its descriptors
	double row_bytes;   // bytes one row's in-list scan is priced at
	double meet_bytes;  // what the pair-centric pre-pass would move for this call (0: it may not run)
	double edge_bytes;  // meet_bias x E (lanes_cost_bytes)
	double bias;
	int mode;           // 1: decide; 2: always
	int64_t V;
};

// Lists the segments of the input: position i starts one when its source differs from row i - 1's or when i is a
// multiple of 1024.  One workgroup per CU takes a contiguous range of windows, counts its starts, reserves their places with
// ONE returning atomic and writes them in order (a wave-level atomic per 64 rows would be 33,000 returning atomics on one
// address for 2.1 M rows: ~0.5 ms; the first version, 1024 workgroups + a ticket: 44 us).  k_src_ball's workgroups read the
// two totals and decide for themselves whether the kernel takes the call (ball_decides).
// seg_rows (a power of two <= 1024): a segment also ends at every multiple of it — 1024 for large calls; a chunk-sized call is
// cut into 128-row segments so that its one or two sources are answered by 16 workgroups at once, each building the same
// (small) ball, instead of by one or two working through 1024 rows each.
__global__ __launch_bounds__(1024) void k_ball_segments(int64_t n, const int64_t *__restrict__ src, u32 *__restrict__ segs,
                                                        MeetDevBlock *__restrict__ db, u32 seg_rows) {
	__shared__ u32 s_cnt[16], s_run[16], s_base;
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	const int64_t nwin = (n + kBallRows - 1) / kBallRows;
	const int64_t per = (nwin + gridDim.x - 1) / gridDim.x;
	const int64_t w0 = (int64_t)blockIdx.x * per, w1 = min(nwin, w0 + per);
	u32 starts = 0, runs = 0;
#pragma unroll 4
	for (int64_t w = w0; w < w1; w++) {
		const int64_t i = w * kBallRows + tid;
		if (i < n) {
			const bool ch = i == 0 || src[i] != src[i - 1];
			runs += ch ? 1u : 0u;
			starts += (ch || ((u32)tid & (seg_rows - 1u)) == 0u) ? 1u : 0u;
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		starts += __shfl_xor(starts, o);
		runs += __shfl_xor(runs, o);
	}
	if (lane == 0) {
		s_cnt[wib] = starts;
		s_run[wib] = runs;
	}
	__syncthreads();
	if (tid == 0) {
		u32 a = 0, b = 0;
		for (int k = 0; k < 16; k++) {
			a += s_cnt[k];
			b += s_run[k];
		}
		s_base = a ? atomicAdd(&db->ball.nseg, a) : 0u;
		if (b) atomicAdd(&db->ball.nrun, b);
	}
	__syncthreads();
	u32 pos = s_base;
	for (int64_t w = w0; w < w1; w++) {
		const int64_t i = w * kBallRows + tid;
		bool st = false;
		if (i < n) st = ((u32)tid & (seg_rows - 1u)) == 0u || src[i] != src[i - 1];
		const u64 m = __ballot(st);
		__syncthreads(); // s_cnt of the window before has been read
		if (lane == 0) s_cnt[wib] = (u32)__popcll(m);
		__syncthreads();
		u32 before = 0, total = 0;
		for (int k = 0; k < 16; k++) {
			const u32 c = s_cnt[k];
			before += k < wib ? c : 0u;
			total += c;
		}
		if (st) segs[pos + before + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)i;
		pos += total;
	}
}

// One ball per segment + one in-list scan per row, against the cheaper of the pre-pass (its measured bytes per row) and the
// lane batches (their level bytes for as many lanes as the input has source runs).  Long segments are bound by the bytes
// they move (2048 x 1024 rows on the SF100-shaped graph: 1.6 GB of 128-byte lines at 4.5 TB/s, whatever the grid); short
// ones by their ~15 dependent round trips on one of the chip's 512 workgroup slots (2048 x 32 rows: 57 us per segment,
// 0.23 ms where the bytes would take 0.07) — in bytes at streaming rate a segment costs at least rule.seg_floor (~512 KB),
// which sends a call of scattered pairs, or of a few rows per source, to the pre-pass.
__device__ __forceinline__ bool ball_decides(const BallRule &rule, int64_t n, u32 nseg, u32 nrun) {
	if (rule.mode == 2) return true;
	const double ball_bytes = fmax((double)nseg * rule.seg_floor, (double)nseg * rule.seg_bytes + (double)n * rule.row_bytes);
	double alt = lanes_cost_bytes(rule.edge_bytes, fmin((double)nrun, (double)rule.V), (double)n, (double)rule.V);
	if (rule.meet_bytes > 0.0) alt = fmin(alt, rule.meet_bytes);
	return rule.bias * ball_bytes <= alt;
}

// One workgroup per segment (persistent grid, segments drawn from a counter).  GM: the bit map is this workgroup's slice of
// a global buffer (V too large for LDS).  qopen: where the rows it cannot answer go (endpoints + row index, one region of
// the pre-pass's queues: when this kernel runs, the stage kernels behind it do not).
// Register discipline: the kernel is compiled for two workgroups per CU (64 VGPRs), and a segment is a dozen short phases
// between barriers.  The first version kept every row's state in registers across the phases: 37 spilled VGPRs, reloaded
// from scratch at the start of nearly every phase — a memory round trip per phase, ~4 us each under load, half of a
// segment's 90 us (option meet_trace).  Now a row's state lives in LDS (result byte, in-list position, descriptor
// position) and a thread carries its destination id alone; the statistics are summed per phase into LDS.
#ifndef PGQ_BALL_WAVES
#define PGQ_BALL_WAVES 8 // wavefronts per SIMD k_src_ball is compiled for: 8 = two 1024-thread workgroups per CU (64 VGPRs), 4 = one (128)
#endif
constexpr int kBallMaxS1 = 4096;  // out-degree up to which a source's two-hop ball is walked (64 rounds of 64 neighbours)
constexpr int kBallFarRows = 16; // rows per segment whose distance-4 walk all 16 wavefronts take up together (k_src_ball)
#ifndef PGQ_BALL_UH
#define PGQ_BALL_UH 4 // row octets (8 rows, 8 lanes each) a wavefront scans per step out of the fixed-stride heads
#endif
#ifndef PGQ_BALL_UQ
#define PGQ_BALL_UQ 2 // row quads (4 rows, 16 lanes each) a wavefront scans per step, all their loads in flight together
#endif
template <bool GM, bool TRACE>
__global__ __launch_bounds__(kBallRows, PGQ_BALL_WAVES) void k_src_ball(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                           int64_t V, const int64_t *__restrict__ off, const int64_t *__restrict__ roff,
                                                           const uint4 *__restrict__ fdesc, const uint4 *__restrict__ rdesc,
                                                           const int32_t *__restrict__ padj, const int32_t *__restrict__ rpadj,
                                                           const uint2 *__restrict__ rseg, const uint4 *__restrict__ rhead,
                                                           const u32 *__restrict__ segs,
                                                           int64_t *__restrict__ out, int64_t cap, int64_t test_cap, int bm_words,
                                                           MeetDevBlock *__restrict__ db, u32 *__restrict__ gmaps, MeetQueue qopen, BallRule rule,
                                                           unsigned long long *__restrict__ trace, u32 seg_rows) {
	extern __shared__ __attribute__((aligned(16))) u32 s_map[]; // bm_words: one bit per vertex (GM: unused)
	const u32 nseg = db->ball.nseg;
	// every workgroup takes the same decision from the same two totals; workgroup 0 leaves it for the kernels behind this one
	const bool go = ball_decides(rule, n, nseg, db->ball.nrun);
	if (blockIdx.x == 0 && threadIdx.x == 0) db->ball.go = go ? 1u : 0u;
	if (!go) return;
	__shared__ signed char s_res[kBallRows]; // per row: its answer, or kBallOpenI
	__shared__ u32 s_d[kBallRows];           // the row's destination (the few rows at distance >= 4 look their descriptors up by it)
	__shared__ uint2 s_in[kBallRows];        // {first group, entries} of the padded in-list of the row's destination
	__shared__ unsigned short s_q1[kBallRows], s_q2[kBallRows], s_q3[kBallRows], s_q4[kBallRows];
	__shared__ u32 s_n1, s_n2, s_n3, s_n4, s_n5, s_len, s_job, s_capped;
	__shared__ int s_flag;
	__shared__ __attribute__((aligned(16))) unsigned char s_win[kBallRows / 64][64];
	__shared__ unsigned long long s_stat[2]; // adjacency entries requested / slot descriptors read by this workgroup
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	const bool head = rhead != nullptr; // the fixed-stride in-list heads exist: a row's scan needs its destination id alone
	unsigned char *win = s_win[wib];
	win[lane] = 0;
	if (tid < 2) s_stat[tid] = 0;
	u32 *const gmap = GM ? gmaps + (size_t)blockIdx.x * bm_words : nullptr;
	auto bit = [&](u32 x) {
		const u32 w = GM ? __hip_atomic_load(&gmap[x >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s_map[x >> 5];
		return (w >> (x & 31)) & 1u;
	};
	auto bits4 = [&](const int4 &v) { return bit((u32)v.x) | bit((u32)v.y) | bit((u32)v.z) | bit((u32)v.w); };
	auto mark = [&](u32 x) {
		if constexpr (GM) { // look first: most entries of a two-hop walk on a skewed graph are the same few hubs (pgq_meet.hip, k_meet4d)
			const u32 w = __hip_atomic_load(&gmap[x >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!((w >> (x & 31)) & 1u)) atomicOr(&gmap[x >> 5], 1u << (x & 31));
		} else {
			atomicOr(&s_map[x >> 5], 1u << (x & 31));
		}
	};
	// option meet_trace: time per phase (10-ns ticks of the constant clock, thread 0 of every workgroup), summed over the grid
	unsigned long long t_last = TRACE ? wall_clock64() : 0ull, t_ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_pmax[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_seg_max = 0, t_seg0 = 0;
	auto tick = [&](int k) {
		if constexpr (TRACE) {
			if (tid == 0) {
				const unsigned long long now = wall_clock64();
				t_ph[k] += now - t_last;
				t_pmax[k] = max(t_pmax[k], now - t_last);
				t_last = now;
			}
		}
	};
	u32 job = blockIdx.x;
	for (;;) {
		__syncthreads(); // the segment before: its LDS state is no longer read
		if (job >= nseg) break;
		tick(7);
		if constexpr (TRACE) t_seg0 = t_last;
		if (tid == 0) {
			s_job = gridDim.x + atomicAdd(&db->ball.next_job, 1u); // read at the segment's end: the round trip is off its path (static striding instead: 0.292 -> 0.302 ms)
			s_len = kBallRows;
			s_n1 = 0;
			s_n2 = 0;
			s_n3 = 0;
			s_n4 = 0;
			s_n5 = 0;
			s_capped = 0;
		}
		const u32 start = segs[job];
		const u32 wend = (u32)min(n, (int64_t)(start | (seg_rows - 1u)) + 1);
		const int64_t s = src[start]; // one address for the whole workgroup
		u32 di32;
		{
			const int64_t ic = min((int64_t)start + tid, n - 1);
			const int64_t si = src[ic], di = dst[ic];
			if constexpr (GM) {
				uint4 *m4 = reinterpret_cast<uint4 *>(gmap);
				for (int k = tid; k < bm_words / 4; k += kBallRows) m4[k] = make_uint4(0, 0, 0, 0);
			} else {
				uint4_alias *m4 = reinterpret_cast<uint4_alias *>(s_map);
				for (int k = tid; k < bm_words / 4; k += kBallRows) m4[k] = make_uint4(0, 0, 0, 0);
			}
			__syncthreads();
			if (start + (u32)tid >= wend || si != s) atomicMin(&s_len, (u32)tid);
			di32 = (di < 0 || di >= V) ? 0xFFFFFFFFu : (u32)di; // out of range: no vertex id (V < 2^31 - 1)
		}
		__syncthreads();
		tick(0);
		const bool sval = s >= 0 && s < V;
		u32 so = 0;
		int degS = 0;
		if (sval) {
			const int64_t a = off[s], b = off[s + 1];
			so = (u32)a;
			degS = (int)(b - a);
		}
		{
			int res = -1; // threads past the segment's end hold no row: closed, so that no later phase takes them up
			uint2 in = make_uint2(0u, 0u);
			if ((u32)tid < s_len) { // >= 1: row `start` matches itself
				res = kBallOpenI;
				if (s < 0) {
					res = -1; // NULL row (iterativelength.cpp:99-101)
				} else if (!sval || di32 == 0xFFFFFFFFu) {
					res = -1;
					db->m.bad = 1;
				} else if ((int64_t)di32 == s) {
					res = 0; // iterativelength.cpp:102-103
				} else if (degS == 0) {
					res = -1; // no path can exist: NULL like an exhausted search (iterativelength.cpp:133-139)
				} else if (!head) {
					in = rseg[di32]; // ONE 8-byte gather per row (a second one for roff[dst] was another 128-byte line per row: a sixth of the kernel's traffic)
					if (in.y == 0) res = -1; // nothing points at dst
				} // (head: the in-degree arrives with the list's head in the scan; a destination nothing points at is closed there)
			}
			s_res[tid] = (signed char)res;
			s_d[tid] = di32;
			s_in[tid] = in;
		}
		const bool need_ball = sval && degS > 0; // workgroup-uniform
		if (need_ball) {
			// S1 = N_out(s)
			for (int k = tid; k < degS; k += kBallRows) mark(fdesc[so + k].x);
			if (tid == 0) atomicAdd(&s_stat[1], 2ull * (unsigned long long)degS); // once as ids, once as the walk's descriptors
			__syncthreads();
			tick(1);
			if (s_res[tid] == kBallOpenI && bit(di32)) s_res[tid] = 1;
			__syncthreads(); // every test against S1 is done before S2's marks land
			// S2 = S1 + N_out(S1): the 16 wavefronts share every round of the source's descriptors.  Not for a hub: its walk is a
			// round of 64 neighbours after the other, each a dependent descriptor read before its few list requests (R-MAT-22:
			// a source with ~100,000 neighbours of ~16 entries each took 5.4 ms over a thousand rounds to reach the cap, the
			// whole kernel's duration — for a ball that is cut, i.e. proves nothing beyond its set bits).  Such a ball stays S1.
			if (degS > kBallMaxS1) {
				if (tid == 0) s_capped = 1;
			} else {
				bool capped = false;
				int resume = 0;
				const unsigned long long e1 = seg_walk<2, false>(
				    fdesc + so, degS, wib, kBallRows / 64, padj, win, false, make_uint4(0, 0, 0, 0),
				    (unsigned long long)cap / (kBallRows / 64), capped, resume,
				    [&](const int4 &v, bool, u32) {
					    mark((u32)v.x);
					    mark((u32)v.y);
					    mark((u32)v.z);
					    mark((u32)v.w);
				    },
				    []() { return false; });
				if (lane == 0 && e1) atomicAdd(&s_stat[0], e1);
				if (capped) s_capped = 1;
			}
			__syncthreads();
			tick(2);
			{
				// S2 incomplete (over the cap): a set bit still proves distance 2 (S1 is complete), nothing else holds
				bool scan = false;
				if (s_res[tid] == kBallOpenI) {
					if (bit(di32)) s_res[tid] = 2;
					else scan = s_capped == 0;
				}
				const u64 m = __ballot(scan);
				u32 base = 0;
				if (lane == 0 && m) base = atomicAdd(&s_n1, (u32)__popcll(m));
				base = (u32)__builtin_amdgcn_readfirstlane((int)base);
				if (scan) s_q1[base + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
			}
			__syncthreads();
			tick(3);
			// distance 3: the in-list of the destination against S2, 16 lanes per row, a wavefront takes 4 x UQ rows per step with
			// all their loads in flight before the first is tested.  The kernel is bound by the bytes it moves (PMC: 1.6 GB per
			// 2.1 M rows at 4.5 TB/s of 128-byte lines), so a list is not read further than needed: (1) its first 64 entries — one
			// 16-byte group per lane, two lines; a destination at distance 3 nearly always shows a witness there; (2) the rest of
			// the lists that showed none, 128 entries per step, compacted into a queue of their own so that the pass is dense.
			{
				constexpr int UQ = PGQ_BALL_UQ;
				const int sub = lane >> 4, j = lane & 15;
				const u32 n1 = s_n1;
				u32 ent = 0;
				if (head) {
					// The list's head sits at rhead[16 d ..): entries 0..30 + the in-degree in the first 128-byte line, entries 31..62
					// in the second.  EIGHT lanes per row and step = one line: the PMC counters of the 16-lane version said the
					// kernel is bound by what it ISSUES (58 M VALU + 38 M SALU + 9 M LDS wave-instructions per 2.1 M rows, phases
					// serialised by barriers), not by its bytes — and a destination at distance 3 shows a witness among its first 31
					// in-neighbours four times out of five.  (a) first line, every row; (b) second line, rows without a witness and
					// more than 31 in-neighbours; (c) the rest of lists longer than 62 (position looked up now), 128 entries per step.
					constexpr int UH = PGQ_BALL_UH; // row octets per wavefront and step: 32 rows, four loads per lane in flight (PGQ_BALL_UH = 4)
					const int sub8 = lane >> 3, j8 = lane & 7;
					for (int pass = 0; pass < 2; pass++) {
						const unsigned short *qin = pass == 0 ? s_q1 : s_q3;
						const u32 nin = pass == 0 ? n1 : s_n4;
						for (u32 base = (u32)wib * (8u * UH); base < nin; base += (u32)(kBallRows / 64) * (8u * UH)) {
							u32 t[UH];
							uint4 x0[UH];
#pragma unroll
							for (int u = 0; u < UH; u++) {
								const u32 it = base + (u32)(8 * u + sub8);
								t[u] = it < nin ? (u32)qin[it] : 0xFFFFFFFFu;
								const u32 d = t[u] != 0xFFFFFFFFu ? s_d[t[u]] : 0u; // idle octets re-read vertex 0's head
								const pgq_v4u r = __builtin_nontemporal_load(reinterpret_cast<const pgq_v4u *>(rhead + (size_t)d * 16 + (u32)(8 * pass + j8)));
								x0[u] = make_uint4(r.x, r.y, r.z, r.w);
							}
#pragma unroll
							for (int u = 0; u < UH; u++) {
								const bool have = t[u] != 0xFFFFFFFFu;
								// the in-degree: word 31 = the first line's last word (pass 1 kept it in s_in)
								const u32 cnt = pass == 0 ? (u32)__shfl((int)x0[u].w, sub8 * 8 + 7) : (have ? s_in[t[u]].y : 0u);
								bool hit = false;
								if (have && cnt > 0) hit = (bit(x0[u].x) | bit(x0[u].y) | bit(x0[u].z) | ((pass == 1 || j8 < 7) ? bit(x0[u].w) : 0u)) != 0;
								const u64 m = __ballot(hit);
								const bool found = ((m >> (8 * sub8)) & 0xFFull) != 0;
								if (have && j8 == 0) {
									if (pass == 0) {
										ent += min(cnt, 31u);
										s_in[t[u]] = make_uint2(0u, cnt); // (the later passes and the distance-4 walk want the in-degree)
										if (cnt == 0) s_res[t[u]] = -1; // nothing points at dst
										else if (found) s_res[t[u]] = 3;
										else if (cnt > 31u) s_q3[atomicAdd(&s_n4, 1u)] = (unsigned short)t[u];
										else s_q2[atomicAdd(&s_n2, 1u)] = (unsigned short)t[u];
									} else {
										ent += min(cnt - 31u, 32u);
										if (found) s_res[t[u]] = 3;
										else if (cnt > 63u) s_q4[atomicAdd(&s_n5, 1u)] = (unsigned short)t[u];
										else s_q2[atomicAdd(&s_n2, 1u)] = (unsigned short)t[u];
									}
								}
							}
						}
						__syncthreads(); // pass 0's queue of second lines is complete (pass 1: so is the queue of long lists)
					}
				} else
				for (u32 base = (u32)wib * (4u * UQ); base < n1; base += (u32)(kBallRows / 64) * (4u * UQ)) {
					u32 t[UQ], ng[UQ];
					uint2 sg[UQ];
					int4 x0[UQ];
#pragma unroll
					for (int u = 0; u < UQ; u++) {
						const u32 it = base + (u32)(4 * u + sub);
						const bool have = it < n1;
						t[u] = have ? (u32)s_q1[it] : 0xFFFFFFFFu;
						sg[u] = have ? s_in[t[u]] : make_uint2(0u, 0u);
						ng[u] = (sg[u].y + 3u) >> 2;
					}
#pragma unroll
					for (int u = 0; u < UQ; u++) // unconditional loads (a load under a per-lane condition is waited for inside its branch): idle lanes re-read the list's first group
						x0[u] = load_group_nt(rpadj, sg[u].x + ((u32)j < ng[u] ? (u32)j : 0u));
#pragma unroll
					for (int u = 0; u < UQ; u++) {
						const bool have = t[u] != 0xFFFFFFFFu;
						const bool hit = (u32)j < ng[u] && bits4(x0[u]);
						const u64 m = __ballot(hit);
						const bool found = ((m >> (16 * sub)) & 0xFFFFull) != 0;
						if (have && j == 0) {
							ent += min(sg[u].y, 64u);
							if (found) s_res[t[u]] = 3;
							else if (ng[u] > 16u) s_q3[atomicAdd(&s_n4, 1u)] = (unsigned short)t[u];
							else s_q2[atomicAdd(&s_n2, 1u)] = (unsigned short)t[u];
						}
					}
				}
				if (!head) __syncthreads();
				const u32 n4 = head ? s_n5 : s_n4;
				const unsigned short *qlong = head ? s_q4 : s_q3;
				for (u32 base = (u32)wib * 4u; base < n4; base += (u32)(kBallRows / 64) * 4u) {
					const u32 it = base + (u32)sub;
					const bool have = it < n4;
					const u32 t = have ? (u32)qlong[it] : 0u;
					// head: the list's position is looked up only now, for the few rows whose first 63 entries showed no witness
					const uint2 sg = have ? (head ? rseg[s_d[t]] : s_in[t]) : make_uint2(0u, 0u);
					const u32 ng = (sg.y + 3u) >> 2;
					bool found = false;
					u32 g0 = head ? 15u : 16u; // (the head holds entries 0..62: group 15 = entries 60..63 is read again)
					while (__any(have && !found && g0 < ng)) {
						const bool act = have && !found && g0 < ng;
						const u32 ga = g0 + (u32)j, gb = ga + 16u;
						const bool a0 = act && ga < ng, a1 = act && gb < ng;
						const int4 y0 = load_group_nt(rpadj, sg.x + (a0 ? ga : 0u));
						const int4 y1 = load_group_nt(rpadj, sg.x + (a1 ? gb : 0u));
						const bool h2 = (a0 && bits4(y0)) || (a1 && bits4(y1));
						const u64 m = __ballot(h2);
						if ((m >> (16 * sub)) & 0xFFFFull) found = true;
						if (act && j == 0) ent += min(sg.y - g0 * 4u, 128u);
						g0 += 32u;
					}
					if (have && j == 0) {
						if (found) s_res[t] = 3;
						else s_q2[atomicAdd(&s_n2, 1u)] = (unsigned short)t;
					}
				}
				for (int o = 32; o > 0; o >>= 1) ent += __shfl_xor(ent, o);
				if (lane == 0 && ent) atomicAdd(&s_stat[0], (unsigned long long)ent);
			}
			__syncthreads();
			tick(4);
			// distance 4: the in-lists of the destination's in-neighbours against S2.  A row at distance 4 has in-neighbours at
			// distance 3, each with a witness in its list, so the first request (256 entries of the first lists) nearly always
			// ends it: (a) one wavefront per row looks at that request alone; (b) what it does not settle — distance >= 5,
			// unreachable, hub destinations — is walked by all 16 wavefronts together, row after row, up to the cap (one
			// wavefront walking 32,768 entries alone held its whole workgroup at the barrier for ~130 us: the first version's tail).
			{
				constexpr int kOneShot = 1 << 20; // request stride that leaves the wavefront exactly one request (no refill)
				const u32 n2 = s_n2;
				for (u32 it = (u32)wib; it < n2; it += (u32)(kBallRows / 64)) {
					const u32 t = (u32)s_q2[it];
					const int degD = (int)s_in[t].y;
					const uint4 *dl = rdesc + roff[s_d[t]];
					uint4 d0 = make_uint4(0, 0, 0, 0);
					if (lane < degD) d0 = dl[lane];
					bool f = false, capped = false;
					int resume = 0;
					const unsigned long long e2 = seg_walk<1, false>(
					    dl, min(degD, 64), 0, kOneShot, rpadj, win, true, d0, 0ull, capped, resume,
					    [&](const int4 &v, bool, u32) { f |= bits4(v) != 0; }, []() { return true; });
					const bool any_f = __any(f) != 0;
					if (lane == 0) {
						atomicAdd(&s_stat[0], e2);
						atomicAdd(&s_stat[1], (unsigned long long)min(degD, 64));
						if (any_f) s_res[t] = 4;
						else s_q1[atomicAdd(&s_n3, 1u)] = (unsigned short)t; // (the scan's queue is done with: the barrier above)
					}
				}
				__syncthreads();
				// ... at most kBallFarRows of them: a source with a small ball and many far or unreachable destinations (R-MAT-22: one
				// segment walked its 32 K-entry allowance for several hundred rows, 6.5-8.9 ms of a 9.4-ms kernel) leaves the rest
				// open — the pair-centric kernels behind take them a row per workgroup, side by side
				const u32 n3 = min(s_n3, (u32)kBallFarRows);
				for (u32 r = 0; r < n3; r++) {
					const u32 t = (u32)s_q1[r];
					const int degD = (int)s_in[t].y;
					const uint4 *dl = rdesc + roff[s_d[t]];
					if (tid == 0) s_flag = 0;
					__syncthreads();
					bool f = false, capped = false;
					int resume = 0;
					const unsigned long long e3 = seg_walk<2, false>(
					    dl, degD, wib, kBallRows / 64, rpadj, win, false, make_uint4(0, 0, 0, 0),
					    (unsigned long long)test_cap / (kBallRows / 64), capped, resume,
					    [&](const int4 &v, bool, u32) { f |= bits4(v) != 0; },
					    [&]() {
						    if (__any(f)) s_flag = 1;
						    return *(volatile int *)&s_flag != 0;
					    });
					if (__any(f)) s_flag = 1;
					if (lane == 0 && e3) atomicAdd(&s_stat[0], e3);
					__syncthreads();
					if (tid == 0) {
						atomicAdd(&s_stat[1], (unsigned long long)degD);
						if (s_flag) s_res[t] = 4;
					}
				}
			}
			__syncthreads();
			tick(5);
		} else {
			__syncthreads();
		}
		// results; what is still open goes to the queue
		{
			const int r = s_res[tid];
			const bool mine = (u32)tid < s_len;
			const bool open = mine && r == kBallOpenI;
			if (mine) out[(int64_t)start + tid] = open ? kMeetOpen : (int64_t)r;
			const u64 m = __ballot(open);
			if (m) {
				u32 base = 0;
				if (lane == 0) base = atomicAdd(&db->ball.open, (u32)__popcll(m));
				base = (u32)__builtin_amdgcn_readfirstlane((int)base);
				if (open) {
					const u32 p = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
					qopen.src[p] = s;
					qopen.dst[p] = (int64_t)di32;
					qopen.idx[p] = start + (u32)tid;
				}
			}
		}
		__syncthreads();
		tick(6);
		if constexpr (TRACE) t_seg_max = max(t_seg_max, t_last - t_seg0);
		job = s_job;
	}
	if constexpr (TRACE) {
		if (tid == 0) {
			for (int k = 0; k < 8; k++) atomicAdd(&trace[k], t_ph[k]);
			atomicMax(&trace[8], t_seg_max);
			for (int k = 0; k < 8; k++) atomicMax(&trace[9 + k], t_pmax[k]); // the longest single phase of each kind
		}
	}
	// statistics: one pair of atomics per workgroup, spread over 32 slots
	if (tid == 0) {
		if (s_stat[0]) atomicAdd(&db->ball.entries[blockIdx.x % 32], s_stat[0]);
		if (s_stat[1]) atomicAdd(&db->ball.descs[blockIdx.x % 32], s_stat[1]);
	}
}

} // namespace pgq

// pgq_meet.hip — pair-centric pre-pass of iterativelength: hop counts 1..3 straight from the CSR (k_meet3).
//
// What it computes is exactly what IterativeLengthFunction reports (iterativelength.cpp:34-143): the BFS distance of
// (src, dst), which is a pure function of (CSR, src, dst).  A pair has
//     distance 1  iff  dst in N_out(src)
//     distance 2  iff  N_out(src) and N_in(dst) intersect
//     distance 3  iff  some out-neighbour of N_out(src) lies in N_in(dst)      (forward expansion)
//                 iff  some in-neighbour of N_in(dst) lies in N_out(src)       (backward expansion)
// and the smallest of these that holds is the answer.  On social graphs almost every random pair is that close (SF100
// knows graph: 97 % of the pairs at distance <= 3), and deciding it costs the pair one scan of a two-hop
// neighbourhood — on average 2.2 K adjacency entries until the first witness, from the endpoint with the shorter walk —
// instead of a share of a full-width MS-BFS level over all 39.9 M in-edges.  The chain:
//   * k_meet3, one wavefront per pair: the one-hop list of the side that is NOT expanded (<= 512 ids) sits in registers
//     behind a 4-KB two-bit filter in LDS; the other side's one-hop list arrives as slot descriptors and its lists are
//     walked as one virtual sequence of 16-byte groups (seg_walk, pgq_walk.h), ended by the first pass with a witness;
//   * k_meet4d / k_meet4 (paths), one 1024-thread workgroup per row k_meet3 leaves open, an exact vertex bit map as the
//     set: distance <= 4;
//   * k_bibfs, one bidirectional BFS per row for a handful of leftovers;
//   * every kernel appends the rows it cannot answer to the next one's queue, the last workgroup of the last kernel
//     reports into pinned host memory: the host launches 2-4 kernels and waits once.
// Rows still open at the end (far apart, unreachable, over the caps) are left to the lane-batched MS-BFS.  Bound: HBM
// (segmented streaming of adjacency entries); algorithmic bytes = 4 B per adjacency entry scanned + 16 B per slot
// descriptor + per row its ids, offsets and result.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>

#include "pgq_search.h"
#include "pgq_walk.h"

namespace pgq {

// what the pre-pass knows about a row's shortest path: its inner vertices, chosen by the reference's tie-break
// (shortest_path.cpp:21-31: parent = smallest vertex of the previous level with an edge to the child)
struct MeetPath {
	int32_t v1, v2, v3, pad;
};

constexpr int kMeet3ExpMax = 4096;  // longest one-hop list k_meet3 expands from (one wavefront); beyond: k_meet4d
constexpr int kMeetWPB = 1;         // wavefronts per k_meet3 workgroup: one, so that a finished row frees its slot at once
#ifndef PGQ_MEET3_WAVES
#define PGQ_MEET3_WAVES 8 // wavefronts per SIMD k_meet3<false> is compiled for (64 VGPRs; 4.2 KB of LDS per wavefront: 8 fit)
#endif
#ifndef PGQ_MEET3_DEPTH
#define PGQ_MEET3_DEPTH 2 // list requests in flight per wavefront
#endif
#ifndef PGQ_MEET3_DEPTH_SMALL
#define PGQ_MEET3_DEPTH_SMALL 4 // ... in the variant for calls too small to fill the chip (latency, not bandwidth, is what counts)
#endif
#ifndef PGQ_MEET4_WAVES
#define PGQ_MEET4_WAVES 16 // wavefronts per k_meet4d row (= workgroup)
#endif
constexpr int kM4Threads = 64 * PGQ_MEET4_WAVES;
#ifndef PGQ_MEET4_BLOCKS
#define PGQ_MEET4_BLOCKS 8 // wavefronts per SIMD k_meet4d is compiled for: 8 = two 1024-thread workgroups per CU (64 VGPRs); at 84 VGPRs only one fits
#endif
#ifndef PGQ_MEET4_MARK_DEPTH
#define PGQ_MEET4_MARK_DEPTH 2 // the marking walk of k_meet4d never stops early: deeper costs nothing but registers
#endif
#ifndef PGQ_BIBFS_DEPTH
#define PGQ_BIBFS_DEPTH 2 // list requests in flight per wavefront of k_bibfs (4: R-MAT-22's far rows 0.12 -> 0.15 ms, more read past the meeting point)
#endif
// (a k_meet4d variant with 8 requests in flight per wavefront for small calls — one workgroup per CU has the registers —
// measured no better on R-MAT-22's long rows, which wait for L2 atomics, not for loads, and worse on the SF100 graph)
#ifndef PGQ_MEET4_DEPTH
#define PGQ_MEET4_DEPTH 2 // the same for each of the 16 wavefronts of a k_meet4d row (more only adds overshoot past the first hit)
#endif
constexpr u32 kMeetKnown4Bit = 0x80000000u; // in the queued row indices: the row is known to be at distance >= 4
constexpr int kMeetStatSlots = 256; // statistics are spread over slots: 10^4 atomics on one address take longer than the walks
struct MeetCounters {
	unsigned long long entries[kMeetStatSlots];  // adjacency entries scanned (both kinds of list)
	unsigned long long vertices[kMeetStatSlots]; // vertices expanded (offset pairs fetched)
	u32 bad;                    // an id outside [0, V)
	u32 pad[3];
};

// ---- round 4: the launch chain without compaction kernels ---------------------------------------------------------------
// Round 3's chain was memset, k_meet3, k_collect_open, k_meet4d, k_collect_open, (k_bibfs, k_collect_open,) copy, wait:
// ~0.05 ms of fixed cost per call, most of an 8192-row call and of a DuckDB chunk.  Now every stage kernel APPENDS the
// rows it cannot answer to the next stage's queue itself (one atomic per open row, ~2 % of the rows) together with what
// it already knows about them (MeetEntry: the endpoints, their list positions and lengths — the next stage needs no
// offset look-up — and which distances are excluded), and the last workgroup of the last kernel of the chain sums the
// statistics, writes them straight into the pinned host block and zeroes the device block for the next call: no memset,
// no compaction launch, no copy command.  The queues ping-pong between two regions (stage 1 -> A, stage 2 -> B, ...).
struct MeetEntry { // 48 bytes: three 16-byte loads per row for the kernel that takes it up
	u32 row, flags, s, d;
	u32 so, degS, di, degD; // forward list of s: adj[so .. so + degS); in-list of d: radj[di .. di + degD)
	u32 workS, workD, pad0, pad1; // entries of the forward two-hop walk of s / the backward one of d (pgq_csr::fwork, rwork)
};
constexpr u32 kEntKnown4 = 1u; // distances 1..3 are excluded (k_meet3 walked to the end)
constexpr u32 kEntKnown3 = 2u; // distances 1..2 are excluded and the two-hop walk below was cut at k_meet3's cap
constexpr u32 kEntFwd = 4u;    // ... it walked forward from s (set = in-list of d); else backward from d (set = out-list of s)
constexpr int kEntResumeShift = 8; // ... and was in the round of descriptors starting at flags >> 8 when it was cut
struct MeetQueue {
	int64_t *src, *dst; // the open rows' endpoints (what the lane-batched search and the older kernels read)
	u32 *idx;           // row index | kMeetKnown4Bit
	MeetEntry *ent;
	u32 *count;
	// Two-ended (k_meet3 -> k_meet4d only): rows with a long way to go (cut walks, lists over the register set) are
	// appended from the front, rows proven to be at distance >= 4 from the back (position cap - 1 - k, counted in
	// *count_back), so that the consumer — which hands out positions in ascending order — starts the long rows first:
	// a 40-us row that starts last is 40 us of tail.  count_back == nullptr: one end.
	u32 *count_back;
	u32 cap;
};
// round 6: the source-centric kernels at the head of the chain (pgq_ball.h)
struct BallDev {
	unsigned long long entries[32], descs[32]; // adjacency entries / slot descriptors read, spread over slots
	u32 nseg, nrun; // segments listed (source runs cut at 1024-row windows) and source runs of the input
	u32 go;         // 1: k_src_ball takes the call — the stage kernels behind it return at once
	u32 next_job, open, ticket, pad[2];
};
struct MeetDevBlock { // device side; all zero between calls
	MeetCounters m;
	u32 count[4]; // rows open after stage 1, 2, 3; [3]: stage 1's rows appended from the back of its queue
	u32 ticket;   // workgroups of the chain's last kernel that are done
	u32 next_job; // k_meet4d: the next queue position to hand out
	u32 pad[2];
	MeetDecision dec;
	BallDev ball;
};
struct MeetHostBlock { // pinned host memory, written by the last workgroup of the chain
	unsigned long long entries[3], vertices[3]; // per stage: k_meet3, the bit-map kernel, k_bibfs
	u32 bad, count[3];
	MeetDecision dec;
	u32 done, count_back;
	unsigned long long ball_entries, ball_descs;
	u32 ball_go, ball_nseg, ball_nrun, ball_open;
};
static_assert(sizeof(MeetHostBlock) <= 8192, "pinned statistics block too small");
__device__ __forceinline__ void queue_push(const MeetQueue &q, u32 row, const MeetEntry &e) { // one lane
	u32 p;
	if (q.count_back && (e.flags & kEntKnown4)) p = q.cap - 1u - atomicAdd(q.count_back, 1u);
	else p = atomicAdd(q.count, 1u);
	q.src[p] = (int64_t)e.s;
	q.dst[p] = (int64_t)e.d;
	q.idx[p] = row | ((e.flags & kEntKnown4) ? kMeetKnown4Bit : 0u);
	uint4 *t = reinterpret_cast<uint4 *>(q.ent + p);
	t[0] = make_uint4(e.row, e.flags, e.s, e.d);
	t[1] = make_uint4(e.so, e.degS, e.di, e.degD);
	t[2] = make_uint4(e.workS, e.workD, 0u, 0u);
}
// position of job j of a (possibly two-ended) queue with nf rows at the front
__device__ __forceinline__ u32 queue_pos(const MeetQueue &q, u32 j, u32 nf) { return j < nf ? j : q.cap - 1u - (j - nf); }
// Every workgroup of a stage kernel ends here (all its threads).  fin == nullptr: not the chain's last kernel.  The last
// workgroup to arrive sums the statistic slots, writes the host block (coherent pinned memory: visible to the host once
// the stream has drained) and leaves the device block zeroed.
__device__ __forceinline__ void meet_finalize(MeetDevBlock *db, MeetHostBlock *fin) {
	if (!fin) return;
	__shared__ u32 s_last;
	__shared__ unsigned long long s_sum[6];
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence_system(); // this workgroup's results (possibly in pinned host memory) before its ticket
		s_last = atomicAdd(&db->ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
		for (int k = 0; k < 6; k++) s_sum[k] = 0;
	}
	__syncthreads();
	if (!s_last) return;
	__threadfence();
	// one thread per slot (the first 256 threads of the workgroup; a one-wavefront workgroup loops): the slot's stage is
	// fixed by its index, so the per-stage sums are three LDS accumulators
	for (int k = threadIdx.x; k < kMeetStatSlots; k += blockDim.x) {
		const unsigned long long e = __hip_atomic_load(&db->m.entries[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned long long v = __hip_atomic_load(&db->m.vertices[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		db->m.entries[k] = 0;
		db->m.vertices[k] = 0;
		const int st = k < 192 ? 0 : (k < 224 ? 1 : 2);
		if (e) atomicAdd(&s_sum[2 * st], e);
		if (v) atomicAdd(&s_sum[2 * st + 1], v);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int k = 0; k < 3; k++) {
			fin->entries[k] = s_sum[2 * k];
			fin->vertices[k] = s_sum[2 * k + 1];
		}
		fin->bad = __hip_atomic_load(&db->m.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (int k = 0; k < 3; k++) {
			fin->count[k] = __hip_atomic_load(&db->count[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			db->count[k] = 0;
		}
		fin->dec = db->dec;
		{ // the source-centric kernels' block (they ran, or declined, at the head of this chain)
			unsigned long long be = 0, bd = 0;
			for (int k = 0; k < 32; k++) {
				be += __hip_atomic_load(&db->ball.entries[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				bd += __hip_atomic_load(&db->ball.descs[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				db->ball.entries[k] = 0;
				db->ball.descs[k] = 0;
			}
			fin->ball_entries = be;
			fin->ball_descs = bd;
			fin->ball_go = db->ball.go;
			fin->ball_nseg = db->ball.nseg;
			fin->ball_nrun = db->ball.nrun;
			fin->ball_open = __hip_atomic_load(&db->ball.open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			db->ball.go = 0;
			db->ball.nseg = 0;
			db->ball.nrun = 0;
			db->ball.open = 0;
			db->ball.next_job = 0;
			db->ball.ticket = 0;
		}
		fin->count_back = __hip_atomic_load(&db->count[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		db->m.bad = 0;
		db->count[3] = 0;
		db->ticket = 0;
		db->next_job = 0;
		db->dec.go = 0;
		__threadfence_system();
		fin->done = 1;
	}
}
// per-workgroup statistics: one pair of atomics, spread over the slots of the kernel's stage (k_meet3: 192 slots for up to
// 65,536 workgroups; the bit-map kernels and k_bibfs: 32 each), so that every kernel's algorithmic bytes can be stated
__device__ __forceinline__ void meet_add_stats(MeetCounters *mc, int stage, unsigned long long entries, unsigned long long vertices) {
	const int base = stage == 0 ? 0 : (stage == 1 ? 192 : 224), cnt = stage == 0 ? 192 : 32;
	if (entries) atomicAdd(&mc->entries[base + blockIdx.x % cnt], entries);
	if (vertices) atomicAdd(&mc->vertices[base + blockIdx.x % cnt], vertices);
}

} // namespace pgq
#include "pgq_ball.h"
namespace pgq {

// One wavefront per row.  A row costs ~4 dependent memory round trips before its walk starts (row, offsets, the two
// one-hop lists — the expanded side's arrives as slot descriptors, so the ranges of its vertices need no look-up) and a
// single wavefront streams only a few KB per round trip, so what counts is how many rows a CU has in flight — the
// workgroup is a single wavefront so that a finished row frees its slot at once, and since round 4 it needs 4.2 KB of
// LDS (the filter; the set's ids sit in registers, pgq_walk.h) so that 8 wavefronts fit a SIMD.  The walk itself is
// seg_walk (pgq_walk.h): the padded lists of a round of 64 expanded vertices as one virtual sequence of 16-byte groups,
// every lane of a request useful.  `cap` bounds the entries a row may WALK (nearly every walk ends at its first hit long
// before): a row over it goes to k_meet4d (16 wavefronts), which takes the walk up in the round it was cut in.
// `go` (nullable): device flag written by k_meet_decide; 0 = the host will take the lane-batched path, do nothing.
// PATHS: also record the path's inner vertices (MeetPath); the walk then runs from dst over the source-ordered in-lists.
// BIGV: V > 2^20, the filter folds the higher id bits in.  DEPTH: list requests in flight.
template <bool PATHS, bool BIGV, int DEPTH>
__global__ __launch_bounds__(64 * kMeetWPB, PATHS ? 6 : (DEPTH > 4 ? 4 : (DEPTH > 2 ? 5 : PGQ_MEET3_WAVES))) void k_meet3(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                  int64_t V, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                  const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                                  const uint4 *__restrict__ fdesc, const uint4 *__restrict__ rdesc,
                                                  const int32_t *__restrict__ padj, const int32_t *__restrict__ rpadj,
                                                  const u32 *__restrict__ fwork, const u32 *__restrict__ rwork,
                                                  int64_t *__restrict__ out, MeetPath *__restrict__ rec, int64_t cap,
                                                  const u32 *__restrict__ go, MeetDevBlock *__restrict__ db, MeetQueue q,
                                                  MeetHostBlock *__restrict__ fin) {
	static_assert(kMeetWPB == 1, "one wavefront per workgroup: the LDS arrays are addressed statically");
	__shared__ __attribute__((aligned(16))) u32 bm[kFltWords];
	__shared__ __attribute__((aligned(16))) unsigned char win[64];
	MeetCounters *const mc = &db->m;
	if ((go && *go == 0) || db->ball.go) { // (a scalar load: one word read by every wavefront through the vector path is a hot spot on one L2
		meet_finalize(db, fin); // channel — polling it per row while the decision kernel ran beside this one made k_meet3
		return;                 // 0.28 ms instead of 0.16, and with 256 spread copies the two extra stream operations ate the 12 us)
	}
	const int lane = threadIdx.x & 63;
	win[lane] = 0;
	unsigned long long entries = 0; // wave-uniform
	u32 vertices = 0;
	const int64_t wave0 = (int64_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	const int64_t nwaves = (int64_t)((gridDim.x * blockDim.x) >> 6);
	for (int64_t i = wave0; i < n; i += nwaves) {
#ifdef PGQ_MEET3_ROWTRACE
		const unsigned long long rt0 = wall_clock64();
#endif
		const int64_t s = src[i], d = dst[i];
		{
			if (s < 0) { // NULL row (iterativelength.cpp:99-101)
				if (lane == 0) out[i] = -1;
				continue;
			}
			if (s >= V || d < 0 || d >= V) {
				if (lane == 0) {
					mc->bad = 1;
					out[i] = -1;
				}
				continue;
			}
			if (s == d) { // iterativelength.cpp:102-103
				if (lane == 0) out[i] = 0;
				continue;
			}
		}
		const int so = (int)off[s], se = (int)off[s + 1], di = (int)roff[d], de = (int)roff[d + 1];
		const u32 workS = fwork[s], workD = rwork[d];
		const int degS = se - so, degD = de - di;
		{
			if (degS == 0 || degD == 0) { // no path can exist: NULL like an exhausted search (iterativelength.cpp:133-139)
				if (lane == 0) out[i] = -1;
				continue;
			}
		}
		MeetEntry ent = { (u32)i, 0u, (u32)s, (u32)d, (u32)so, (u32)degS, (u32)di, (u32)degD, workS, workD, 0u, 0u };
		// expand from the endpoint whose two-hop walk is the shorter one (round 4: the sums of the neighbours' list lengths
		// are kept per vertex; the shorter one-hop LIST picked the longer walk for every row that ran into the cap, and the
		// walks are 4 % shorter on average); the other endpoint's list is the set.  fwd: walk N_out(N_out(s)) against the
		// set N_in(d).  PATHS always expands from dst against the set N_out(src): the in-lists are ordered by source (built
		// so at upload), so the walk meets the witnesses in the order of the reference's tie-break (smallest second-to-last
		// vertex first, then the smallest vertex before it) and may stop at the first one
		bool fwd = PATHS ? false : workS <= workD;
		// (handing a row on whenever the preferred set does not fit the registers, instead of walking from the other endpoint,
		// sent 1740 instead of 34 such rows of the 65,536 to the bit-map kernel: 0.244 -> 0.264 ms)
		if (!PATHS && (fwd ? degD : degS) > kSetRegMax) fwd = !fwd;
		const int set_n = fwd ? degD : degS;
		const int exp_n = fwd ? degS : degD;
		{
			// both lists too long for the registers — or the expanded side's own list is a hub's: its distance-2 test alone is one
			// wavefront reading exp_n descriptors 256 per round trip (R-MAT-22: a 100,000-neighbour endpoint among 1024 random pairs
			// held the kernel for 70 us); the bit-map kernel reads it with 16 wavefronts
			if (set_n > kSetRegMax || (!PATHS && exp_n > kMeet3ExpMax)) {
				if (lane == 0) {
					out[i] = kMeetOpen;
					queue_push(q, (u32)i, ent);
				}
				continue;
			}
		}
		const int32_t *set_adj = fwd ? radj + di : adj + so;      // the set side's one-hop list (ids)
		const uint4 *exp_desc = fwd ? fdesc + so : rdesc + di;    // the expanded side's one-hop list (slot descriptors)
		const int32_t *xp = fwd ? padj : rpadj;                   // padded adjacency of the expanded side's direction
		const u32 other = (u32)(fwd ? s : d);                     // distance 1: the set list contains the other endpoint
		// both one-hop lists are requested together: the set's ids into registers, the expanded side's first descriptors
		RegSet R;
		R.rounds = (set_n + 63) >> 6;
#pragma unroll
		for (int k = 0; k < kSetRegs; k++) {
			R.r[k] = kMeetEmpty;
			if (k < R.rounds) { // wave-uniform
				const int p = k * 64 + lane;
				const u32 x = (u32)set_adj[min(p, set_n - 1)];
				R.r[k] = p < set_n ? x : kMeetEmpty;
			}
		}
		uint4 d0 = make_uint4(0, 0, 0, 0);
		if (lane < exp_n) d0 = exp_desc[lane];
#pragma unroll
		for (int k = 0; k < kFltWords / 256; k++) reinterpret_cast<uint4_alias *>(bm)[k * 64 + lane] = make_uint4(0, 0, 0, 0);
		__builtin_amdgcn_wave_barrier();
		bool hit = false;
#pragma unroll
		for (int k = 0; k < kSetRegs; k++) {
			if (k < R.rounds) {
				const u32 x = R.r[k];
				if (x != kMeetEmpty) {
					hit |= x == other;
					atomicOr(&bm[flt_word(x)], flt_mask<BIGV>(x));
				}
			}
		}
		entries += (unsigned long long)set_n;
		vertices += (u32)exp_n; // one 16-byte descriptor per expanded vertex
		__builtin_amdgcn_wave_barrier();
		{
			if (__any(hit)) {
				if (lane == 0) out[i] = 1;
				continue;
			}
			// distance 2: the middle vertex is the smallest common one
			u32 mid = kMeetEmpty; // wave-uniform
			{
				const u32 v = lane < exp_n ? d0.x : kMeetEmpty;
				const u32 pass = v != kMeetEmpty ? (flt_test<BIGV>(bm[flt_word(v)], v) & 1u) : 0u;
				verify_candidates(R, pass, make_int4((int)v, 0, 0, 0), [&](u32 x, int) {
					mid = min(mid, x);
					return !PATHS; // a hop count needs any common neighbour, a path the smallest
				});
			}
			// the rest of a long one-hop list, 256 ids per round trip (a hub's 5000 neighbours were 78 dependent trips of 64)
			for (int pb = 64; pb < exp_n; pb += 256) {
				u32 v[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int p = pb + 64 * u + lane;
					v[u] = exp_desc[min(p, exp_n - 1)].x;
					if (p >= exp_n) v[u] = kMeetEmpty;
				}
				u32 pass = 0;
#pragma unroll
				for (int u = 0; u < 4; u++)
					if (v[u] != kMeetEmpty) pass |= (flt_test<BIGV>(bm[flt_word(v[u])], v[u]) & 1u) << u;
				verify_candidates(R, pass, make_int4((int)v[0], (int)v[1], (int)v[2], (int)v[3]), [&](u32 x, int) {
					mid = min(mid, x);
					return !PATHS;
				});
				if (!PATHS && mid != kMeetEmpty) break;
			}
			if (mid != kMeetEmpty) {
				if (lane == 0) {
					if constexpr (PATHS) rec[i].v1 = (int32_t)mid;
					out[i] = 2;
				}
				continue;
			}
		}
		// distance 3: the padded lists of the expanded side's vertices, one filter probe per entry (the exact test only for
		// what the filter lets through), ended by the first pass with a hit.  PATHS: requests are processed in walk order
		// and the lists ascend, so every later witness has a larger (second vertex, first vertex) key than the smallest one
		// of the pass that found the first
		u64 best = ~0ull; // wave-uniform: smallest (outer vertex << 32 | inner vertex) over the witnesses (PATHS), or 0 = found
		bool capped = false;
		int resume = 0;
#ifdef PGQ_MEET3_ROWTRACE
		const unsigned long long rt1 = wall_clock64();
		const unsigned long long ent_before = entries;
#endif
		entries += seg_walk<DEPTH, PATHS>(
		    exp_desc, exp_n, 0, 1, xp, win, true, d0, (unsigned long long)cap, capped, resume,
		    [&](const int4 &v, bool ok, u32 ev) {
			    // a lane past the round's end re-reads real entries of the last list: no mask needed for membership;
			    // PATHS masks them (their ev is the last list's, so they would even be right, but cost a verification)
			    const u32 pass = (PATHS && !ok) ? 0u : flt_pass4<BIGV>(bm, v);
			    verify_candidates(R, pass, v, [&](u32 x, int L) {
				    if constexpr (PATHS) { // backward walk: expanded vertex = second-to-last, entry = the one before it
					    const u32 y = (u32)__builtin_amdgcn_readlane((int)ev, L);
					    best = min(best, (u64)y << 32 | x);
					    return false;
				    } else {
					    best = 0;
					    return true;
				    }
			    });
		    },
		    [&]() { return best != ~0ull; });
		const bool found = best != ~0ull;
#ifdef PGQ_MEET3_ROWTRACE
		{
			const unsigned long long rt2 = wall_clock64();
			if (lane == 0 && rt2 - rt0 > 2000) // > 20 us
				printf("meet3 row %lld: start +%.1f us, head %.1f us, walk %.1f us, degS %d degD %d workS %u workD %u set %d exp %d walked %llu found %d capped %d\n", (long long)i,
				       0.0, (rt1 - rt0) * 0.01, (rt2 - rt1) * 0.01, degS, degD, workS, workD, set_n, exp_n, entries - ent_before, (int)found, (int)capped);
		}
#endif
		if (lane == 0) {
			if constexpr (PATHS) {
				if (found) {
					rec[i].v2 = (int32_t)(best >> 32);
					rec[i].v1 = (int32_t)(u32)best;
				}
			}
			// the walk ran to its end without a witness: the distance is at least 4; it was cut short: distances 1 and 2
			// are excluded and the next stage takes the walk up where it stopped
			out[i] = found ? 3 : (capped ? kMeetOpen : kMeetOpen4);
			if (!found) {
				ent.flags = capped ? (kEntKnown3 | (fwd ? kEntFwd : 0u) | ((u32)resume << kEntResumeShift)) : kEntKnown4;
				queue_push(q, (u32)i, ent);
			}
		}
	}
	if (lane == 0) meet_add_stats(mc, 0, entries, (unsigned long long)vertices);
	meet_finalize(db, fin);
}

// ---- k_meet3 for chunk-sized calls: several wavefronts per row (k_meet3w, round 6) ------------------------------------
// A call of <= meet_wide_rows rows (one or two DuckDB chunks) leaves most of the chip's 8192 wavefront slots empty, and its
// duration is its slowest row's: a two-hop walk of ~9,000 entries that meets its witness late (or not at all: distance >= 4)
// is ~40 list requests, 10 dependent round trips of one wavefront with four in flight — 18-22 us of k_meet3's 28 (row trace,
// 2048 rows of the SF100-shaped graph).  Here a row is a workgroup of WPB wavefronts: all of them hold the set in their
// registers (the same loads, cached), the filter is built once in LDS, distances 1 and 2 are tested by every wavefront alike
// (so that the branches stay uniform across the workgroup), and the two-hop walk is split request by request (seg_walk's
// stride); the first witness any of them finds ends the row.  Same tests in the same order as k_meet3: same answers.
template <bool BIGV, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_meet3w(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                    int64_t V, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                    const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                                    const uint4 *__restrict__ fdesc, const uint4 *__restrict__ rdesc,
                                                    const int32_t *__restrict__ padj, const int32_t *__restrict__ rpadj,
                                                    const u32 *__restrict__ fwork, const u32 *__restrict__ rwork,
                                                    int64_t *__restrict__ out, int64_t cap, const u32 *__restrict__ go,
                                                    MeetDevBlock *__restrict__ db, MeetQueue q, MeetHostBlock *__restrict__ fin) {
	static_assert(kFltWords / 256 <= 4 && (WPB == 2 || WPB == 4), "the filter is cleared in at most four 1-KB parts");
	__shared__ __attribute__((aligned(16))) u32 bm[kFltWords];
	__shared__ __attribute__((aligned(16))) unsigned char win_all[WPB][64];
	__shared__ u32 s_found, s_capped;
	MeetCounters *const mc = &db->m;
	if ((go && *go == 0) || db->ball.go) {
		meet_finalize(db, fin);
		return;
	}
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	unsigned char *const win = win_all[wib];
	win[lane] = 0;
	unsigned long long entries = 0; // wave-uniform
	u32 vertices = 0;
	// every branch below is taken by all wavefronts of the workgroup alike (it depends on the row alone)
	for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
		__syncthreads(); // the row before: its filter and flags are no longer read
		const int64_t s = src[i], d = dst[i];
		if (s < 0) { // NULL row (iterativelength.cpp:99-101)
			if (tid == 0) out[i] = -1;
			continue;
		}
		if (s >= V || d < 0 || d >= V) {
			if (tid == 0) {
				mc->bad = 1;
				out[i] = -1;
			}
			continue;
		}
		if (s == d) { // iterativelength.cpp:102-103
			if (tid == 0) out[i] = 0;
			continue;
		}
		const int so = (int)off[s], se = (int)off[s + 1], di = (int)roff[d], de = (int)roff[d + 1];
		const u32 workS = fwork[s], workD = rwork[d];
		const int degS = se - so, degD = de - di;
		if (degS == 0 || degD == 0) { // no path can exist: NULL like an exhausted search (iterativelength.cpp:133-139)
			if (tid == 0) out[i] = -1;
			continue;
		}
		MeetEntry ent = { (u32)i, 0u, (u32)s, (u32)d, (u32)so, (u32)degS, (u32)di, (u32)degD, workS, workD, 0u, 0u };
		bool fwd = workS <= workD; // the side k_meet3 picks
		if ((fwd ? degD : degS) > kSetRegMax) fwd = !fwd;
		const int set_n = fwd ? degD : degS;
		const int exp_n = fwd ? degS : degD;
		if (set_n > kSetRegMax || exp_n > kMeet3ExpMax) {
			if (tid == 0) {
				out[i] = kMeetOpen;
				queue_push(q, (u32)i, ent);
			}
			continue;
		}
		const int32_t *set_adj = fwd ? radj + di : adj + so;
		const uint4 *exp_desc = fwd ? fdesc + so : rdesc + di;
		const int32_t *xp = fwd ? padj : rpadj;
		const u32 other = (u32)(fwd ? s : d);
		RegSet R;
		R.rounds = (set_n + 63) >> 6;
#pragma unroll
		for (int k = 0; k < kSetRegs; k++) {
			R.r[k] = kMeetEmpty;
			if (k < R.rounds) {
				const int p = k * 64 + lane;
				const u32 x = (u32)set_adj[min(p, set_n - 1)];
				R.r[k] = p < set_n ? x : kMeetEmpty;
			}
		}
		uint4 d0 = make_uint4(0, 0, 0, 0);
		if (lane < exp_n) d0 = exp_desc[lane];
		for (int k = wib; k < kFltWords / 256; k += WPB) reinterpret_cast<uint4_alias *>(bm)[k * 64 + lane] = make_uint4(0, 0, 0, 0);
		if (tid == 0) {
			s_found = 0;
			s_capped = 0;
		}
		__syncthreads();
		bool hit = false;
#pragma unroll
		for (int k = 0; k < kSetRegs; k++) {
			if (k < R.rounds) {
				const u32 x = R.r[k];
				if (x != kMeetEmpty) {
					hit |= x == other;
					if ((k % WPB) == wib) atomicOr(&bm[flt_word(x)], flt_mask<BIGV>(x)); // the set's registers are shared out for the filter
				}
			}
		}
		if (wib == 0) {
			entries += (unsigned long long)set_n;
			vertices += (u32)exp_n;
		}
		__syncthreads();
		if (__any(hit)) {
			if (tid == 0) out[i] = 1;
			continue;
		}
		u32 mid = kMeetEmpty; // distance 2: any common neighbour
		{
			const u32 v = lane < exp_n ? d0.x : kMeetEmpty;
			const u32 pass = v != kMeetEmpty ? (flt_test<BIGV>(bm[flt_word(v)], v) & 1u) : 0u;
			verify_candidates(R, pass, make_int4((int)v, 0, 0, 0), [&](u32 x, int) {
				mid = x;
				return true;
			});
		}
		for (int pb = 64; pb < exp_n && mid == kMeetEmpty; pb += 256) {
			u32 v[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int p = pb + 64 * u + lane;
				v[u] = exp_desc[min(p, exp_n - 1)].x;
				if (p >= exp_n) v[u] = kMeetEmpty;
			}
			u32 pass = 0;
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (v[u] != kMeetEmpty) pass |= (flt_test<BIGV>(bm[flt_word(v[u])], v[u]) & 1u) << u;
			verify_candidates(R, pass, make_int4((int)v[0], (int)v[1], (int)v[2], (int)v[3]), [&](u32 x, int) {
				mid = x;
				return true;
			});
		}
		if (mid != kMeetEmpty) {
			if (tid == 0) out[i] = 2;
			continue;
		}
		// distance 3: the walk's requests dealt out to the WPB wavefronts; every one may request cap / WPB entries
		bool mine = false, capped = false;
		int resume = 0;
		entries += seg_walk<PGQ_MEET3_DEPTH_SMALL, false>(
		    exp_desc, exp_n, wib, WPB, xp, win, true, d0, (unsigned long long)cap / WPB, capped, resume,
		    [&](const int4 &v, bool, u32) {
			    const u32 pass = flt_pass4<BIGV>(bm, v);
			    verify_candidates(R, pass, v, [&](u32, int) {
				    mine = true;
				    return true;
			    });
		    },
		    [&]() {
			    if (mine) s_found = 1;
			    return *(volatile u32 *)&s_found != 0u;
		    });
		if (mine) s_found = 1;
		if (capped) s_capped = 1;
		__syncthreads();
		if (tid == 0) {
			const bool found = s_found != 0u, cut = s_capped != 0u;
			out[i] = found ? 3 : (cut ? kMeetOpen : kMeetOpen4);
			if (!found) { // (a cut walk is taken up from its start by the bit-map kernel: the wavefronts stopped in different rounds)
				ent.flags = cut ? (kEntKnown3 | (fwd ? kEntFwd : 0u)) : kEntKnown4;
				queue_push(q, (u32)i, ent);
			}
		}
	}
	if (lane == 0) meet_add_stats(mc, 0, entries, (unsigned long long)vertices);
	meet_finalize(db, fin);
}

// ---- distance <= 4 with an exact vertex bit map, path variant (k_meet4) ---------------------------------------------
// For the `shortestpath` rows k_meet3<true> leaves open (distance 4, or lists / walks over its caps): one 1024-thread
// workgroup per row, the bit map in LDS when it fits (V <= ~1.2 M), else a slice of a global buffer (GM):
//     B = N_out(src)                    distance 1 iff dst in B;  distance 2: smallest vertex of N_in(dst) in B
//     distance 3: backward walk (in-lists of N_in(dst)) against B -> smallest (second vertex, first vertex)
//     B += N_out(N_out(src))            distance 4: backward walk against B -> smallest (third, second); first vertex =
//                                       smallest in-neighbour of the second that src points at
// (each test only runs when the smaller distances have been excluded, so the first that holds is the BFS distance).
// The in-lists are ordered by source, so a backward walk meets the witnesses in the order of the reference's tie-break
// (shortest_path.cpp:21-31: the parent of a vertex is the smallest vertex of the previous level) and a wavefront stops
// once its lists are past the best second-to-last vertex any wavefront has published.  The 16 wavefronts split a
// one-hop list round-robin and stream the segments 16 bytes per lane per request.  Rows whose walks exceed `cap`
// entries stay open.  Distance-only rows take k_meet4d below.

template <bool PATHS, bool GM>
__global__ __launch_bounds__(1024) void k_meet4(MeetQueue qin, int64_t V, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                                const uint4 *__restrict__ fdesc, const uint4 *__restrict__ rdesc,
                                                const int32_t *__restrict__ padj, const int32_t *__restrict__ rpadj,
                                                int64_t *__restrict__ out_rows,
                                                MeetPath *__restrict__ rec_rows, int64_t cap, int bm_words,
                                                MeetDevBlock *__restrict__ db, u32 *__restrict__ gmaps, MeetQueue qout,
                                                MeetHostBlock *__restrict__ fin) {
	extern __shared__ u32 s_map[]; // bm_words: one bit per vertex (GM: the map is this workgroup's slice of `gmaps`)
	const int64_t n = (int64_t)*qin.count; // rows left open by the kernel before (counted on the device: no host round trip)
	const int64_t *const src = qin.src, *const dst = qin.dst;
	const u32 *const didx = qin.idx;
	MeetCounters *const mc = &db->m;
	u32 *const gmap = GM ? gmaps + (size_t)blockIdx.x * bm_words : nullptr;
	__shared__ unsigned long long s_best;
	__shared__ __attribute__((aligned(16))) unsigned char s_win[16][64];
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	unsigned char *win = s_win[wib]; // seg_owner's window of this wavefront
	win[lane] = 0;
	unsigned long long entries = 0;
	u32 vertices = 0;
	// GM: bits are set by L2 atomics, so they are read with device-scope atomic loads (a plain load may hit a stale L1 line)
	auto bit = [&](u32 x) {
		const u32 w = GM ? __hip_atomic_load(&gmap[x >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s_map[x >> 5];
		return (w >> (x & 31)) & 1u;
	};
	auto mark = [&](u32 x) {
		if constexpr (GM) {
			// look first: on a skewed graph most entries of a two-hop walk are the same few hubs, and read-modify-writes of
			// one word serialise in the L2 (R-MAT-22: a 200,000-entry marking walk took 270 us) — a set bit needs no atomic
			const u32 w = __hip_atomic_load(&gmap[x >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!((w >> (x & 31)) & 1u)) atomicOr(&gmap[x >> 5], 1u << (x & 31));
		} else {
			atomicOr(&s_map[x >> 5], 1u << (x & 31));
		}
	};
	auto clear_map = [&]() {
		if constexpr (GM) {
			uint4 *m4 = reinterpret_cast<uint4 *>(gmap); // bm_words is a multiple of 4, slices are 16-byte aligned
			for (int k = tid; k < bm_words / 4; k += 1024) m4[k] = make_uint4(0, 0, 0, 0);
		} else {
			for (int k = tid; k < bm_words; k += 1024) s_map[k] = 0;
		}
	};
	for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
		__syncthreads(); // the previous row's flags and map are no longer read
		const int64_t s = src[i], d = dst[i]; // rows left open by k_meet3: ids in range, src != dst, both have edges
		const u32 row = didx[i] & ~kMeetKnown4Bit;
		const int so = (int)off[s], se = (int)off[s + 1], di = (int)roff[d], de = (int)roff[d + 1];
		const int degS = se - so, degD = de - di;
		// rows k_meet3 walked to the end are known to be at distance >= 4: only the two two-hop walks remain (two
		// dependent phases instead of six; their sizes passed k_meet3's cap, which is below this kernel's)
		const bool known4 = (didx[i] & kMeetKnown4Bit) != 0;
		// the sizes of the two two-hop walks came with the row (k_meet3's queue entry; round 3 summed 2 x degree offset
		// pairs per row here)
		const uint4 ew = reinterpret_cast<const uint4 *>(qin.ent + i)[2];
		const int64_t work_f = (int64_t)(u32)__builtin_amdgcn_readfirstlane((int)ew.x);
		const int64_t work_b = (int64_t)(u32)__builtin_amdgcn_readfirstlane((int)ew.y);
		auto leave_open = [&]() { // one thread: the row goes on to the next stage
			out_rows[row] = kMeetOpen;
			const MeetEntry e = { row, 0u, (u32)s, (u32)d, (u32)so, (u32)degS, (u32)di, (u32)degD, (u32)work_f, (u32)work_b, 0u, 0u };
			queue_push(qout, row, e);
		};
		clear_map();
		if (tid == 0) s_best = ~0ull;
		__syncthreads();
		// the smallest vertex of N_in(dst) whose bit is set (distance 2: the middle vertex; distance 3: the second one)
		auto min_in_neighbour_of_dst = [&]() {
			u32 m = kMeetEmpty;
			for (int p = tid; p < degD; p += 1024) {
				const u32 u = (u32)radj[di + p];
				if (bit(u)) m = min(m, u);
			}
			if (m != kMeetEmpty) atomicMin(&s_best, (unsigned long long)m);
		};
		// PATHS: first inner vertex = smallest in-neighbour of `v` that src points at.  The map is rebuilt as N_out(src).
		auto first_inner_vertex = [&](u32 v) -> u32 {
			__syncthreads();
			clear_map();
			if (tid == 0) s_best = ~0ull;
			__syncthreads();
			for (int p = tid; p < degS; p += 1024) {
				const u32 x = (u32)adj[so + p];
				mark(x);
			}
			__syncthreads();
			u32 m = kMeetEmpty;
			const int vb = (int)roff[v], ve = (int)roff[v + 1];
			for (int p = vb + tid; p < ve; p += 1024) {
				const u32 y = (u32)radj[p];
				if (bit(y)) m = min(m, y);
			}
			if (m != kMeetEmpty) atomicMin(&s_best, (unsigned long long)m);
			__syncthreads();
			return (u32)s_best;
		};
		if (tid == 0) {
			entries += (unsigned long long)(degS + degD);
			vertices += (u32)(degS + degD);
		}
		if (!known4) {
			// B = N_out(src)
			for (int p = tid; p < degS; p += 1024) mark((u32)adj[so + p]);
			__syncthreads();
			if (bit((u32)d)) { // dst in N_out(src)
				if (tid == 0) out_rows[row] = 1;
				continue;
			}
			min_in_neighbour_of_dst();
			// every wavefront reads s_best between two barriers: the walks below publish into it again, and a wavefront
			// delayed between a barrier and its read would otherwise take another branch than the rest
			__syncthreads();
			const unsigned long long best2 = s_best;
			__syncthreads();
			if (best2 != ~0ull) {
				if (tid == 0) {
					out_rows[row] = 2;
					if constexpr (PATHS) rec_rows[row].v1 = (int32_t)(u32)best2;
				}
				continue;
			}
			if (work_b > cap) {
				if (tid == 0) leave_open();
				continue;
			}
		}
		// The backward walk: in-lists of N_in(dst) in list order = ascending second-to-last vertex y, entries ascending, so
		// the smallest key (y << 32 | x) over the entries x whose bit is set is the reference's choice (smallest parent at
		// every step back from dst).  A wavefront stops once its lists are past the best y any wavefront has published.
		// Round 4: the walk is seg_walk in cooperative form (every wavefront takes every 16th request of a round of 64 in-lists,
		// pgq_walk.h) over the slot descriptors — no offset look-up per expanded vertex, every lane of a request useful; round
		// 3 dealt whole lists to the wavefronts (meet_walk).  Requests are taken in walk order, so the stop rule holds as
		// before: what a wavefront has not requested yet lies behind the lists it has seen.
		auto backward_walk = [&]() {
			unsigned long long best = ~0ull;
			u32 last = 0;
			bool capped = false;
			int resume = 0;
			const unsigned long long e2 = seg_walk<2, true>(
			    rdesc + di, degD, wib, 16, rpadj, win, false, make_uint4(0, 0, 0, 0), ~0ull, capped, resume,
			    [&](const int4 &v, bool ok, u32 ev) {
				    last = ev; // lanes past the round's end carry the last list's vertex: not smaller than any real one
				    if (ok) {
					    if (bit((u32)v.x)) best = min(best, (unsigned long long)ev << 32 | (u32)v.x);
					    if (bit((u32)v.y)) best = min(best, (unsigned long long)ev << 32 | (u32)v.y);
					    if (bit((u32)v.z)) best = min(best, (unsigned long long)ev << 32 | (u32)v.z);
					    if (bit((u32)v.w)) best = min(best, (unsigned long long)ev << 32 | (u32)v.w);
				    }
			    },
			    [&]() {
				    if (__any(best != ~0ull)) {
					    best = wave_min_u64(best);
					    if (lane == 0) atomicMin(&s_best, best);
				    }
				    const u32 by = (u32)(*(volatile unsigned long long *)&s_best >> 32);
				    return __any(last > by) != 0;
			    });
			if (lane == 0) entries += e2;
			best = wave_min_u64(best);
			if (lane == 0 && best != ~0ull) atomicMin(&s_best, best);
		};
		if (!known4) {
			backward_walk(); // distance 3: against the map of N_out(src)
			__syncthreads();
			const unsigned long long best3 = s_best;
			__syncthreads();
			if (best3 != ~0ull) {
				if (tid == 0) {
					if constexpr (PATHS) {
						rec_rows[row].v1 = (int32_t)(u32)best3;
						rec_rows[row].v2 = (int32_t)(u32)(best3 >> 32);
					}
					out_rows[row] = 3;
				}
				continue;
			}
			if (work_f > cap) {
				if (tid == 0) leave_open();
				continue;
			}
		}
		// B += N_out(N_out(src))
		{
			bool capped = false;
			int resume = 0;
			const unsigned long long e2 = seg_walk<2, false>(
			    fdesc + so, degS, wib, 16, padj, win, false, make_uint4(0, 0, 0, 0), ~0ull, capped, resume,
			    [&](const int4 &v, bool, u32) { // padding and re-read groups repeat real entries: a mark does not mind
				    mark((u32)v.x);
				    mark((u32)v.y);
				    mark((u32)v.z);
				    mark((u32)v.w);
			    },
			    []() { return false; });
			if (lane == 0) entries += e2;
		}
		__syncthreads();
		backward_walk(); // distance 4: third vertex y, second vertex x
		__syncthreads();
		const bool found4 = s_best != ~0ull;
		if (found4) {
			const unsigned long long key = s_best;
			__syncthreads();
			if constexpr (PATHS) {
				const u32 v3 = (u32)(key >> 32), v2 = (u32)key;
				const u32 v1 = first_inner_vertex(v2);
				if (tid == 0) {
					rec_rows[row].v1 = (int32_t)v1;
					rec_rows[row].v2 = (int32_t)v2;
					rec_rows[row].v3 = (int32_t)v3;
				}
			}
		}
		if (tid == 0) {
			if (found4) out_rows[row] = 4;
			else leave_open();
		}
	}
	__shared__ unsigned long long s_stat[2];
	__syncthreads();
	if (tid < 2) s_stat[tid] = 0;
	__syncthreads();
	for (int o = 32; o > 0; o >>= 1) entries += __shfl_xor(entries, o);
	if (lane == 0 && entries) atomicAdd(&s_stat[0], entries);
	if (tid == 0 && vertices) atomicAdd(&s_stat[1], (unsigned long long)vertices);
	__syncthreads();
	if (tid == 0) meet_add_stats(mc, 1, s_stat[0], s_stat[1]);
	meet_finalize(db, fin);
}

// ---- distance only: k_meet4d -------------------------------------------------------------------------------------------
// The rows that reach this kernel from `iterativelength` are of two kinds: rows whose cheaper two-hop walk is over
// k_meet3's cap (heavy endpoints; nearly all of them at distance 2 or 3) and rows k_meet3 walked to the end (distance
// >= 4 proven).  Marking a whole two-hop neighbourhood before testing anything, as the path variant above must (it needs
// the smallest witness), would read a heavy row's 10^5..10^6 entries to answer what the first few thousand already
// decide, so the distance-only flow tests in k_meet3's order with 16 wavefronts per row and the exact bit map as the set:
//     sizes of both two-hop walks, distance 1                                  (the one-hop lists)
//     map = one-hop list of the endpoint with the LARGER walk; distance 2      (the other endpoint's one-hop list)
//     distance 3: two-hop walk of the cheaper endpoint against the map, ended by the first hit
//     distance 4: map = two-hop set of the cheaper endpoint, two-hop walk of the other one, ended by the first hit
// Rows with distance >= 4 proven start at the last step (cheaper endpoint: the shorter one-hop list).
template <bool GM, bool TRACE>
__global__ __launch_bounds__(64 * PGQ_MEET4_WAVES, PGQ_MEET4_BLOCKS) void k_meet4d(MeetQueue qin, const int32_t *__restrict__ adj, const int32_t *__restrict__ radj,
                                                 const uint4 *__restrict__ fdesc, const uint4 *__restrict__ rdesc,
                                                 const int32_t *__restrict__ padj, const int32_t *__restrict__ rpadj,
                                                 int64_t *__restrict__ out_rows, int64_t cap, int64_t test_cap, int bm_words,
                                                 MeetDevBlock *__restrict__ db, u32 *__restrict__ gmaps, MeetQueue qout,
                                                 MeetHostBlock *__restrict__ fin, unsigned long long *__restrict__ trace, SampleArgs sm) {
	extern __shared__ __attribute__((aligned(16))) u32 s_map[]; // bm_words: one bit per vertex (GM: the map is this workgroup's slice of `gmaps`)
	// rows the stage before left open (counted on the device: no host round trip): nf from the front of its queue (the
	// long ones), the rest from the back
	const u32 nf = *qin.count, n = nf + (qin.count_back ? *qin.count_back : 0u);
	u32 *const gmap = GM ? gmaps + (size_t)blockIdx.x * bm_words : nullptr;
	__shared__ int s_flag;
	__shared__ int s_capped;
	__shared__ u32 s_job;
	__shared__ __attribute__((aligned(16))) unsigned char s_win[PGQ_MEET4_WAVES][64];
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	unsigned char *win = s_win[wib]; // seg_owner's window of this wavefront
	win[lane] = 0;
	unsigned long long entries = 0; // wave-uniform: this wavefront's share
	u32 vertices = 0;               // wave-uniform (wavefront 0 counts)
	auto bit = [&](u32 x) {
		const u32 w = GM ? __hip_atomic_load(&gmap[x >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s_map[x >> 5];
		return (w >> (x & 31)) & 1u;
	};
	auto mark = [&](u32 x) {
		if constexpr (GM) {
			// look first: on a skewed graph most entries of a two-hop walk are the same few hubs, and read-modify-writes of
			// one word serialise in the L2 (R-MAT-22: a 200,000-entry marking walk took 270 us) — a set bit needs no atomic
			const u32 w = __hip_atomic_load(&gmap[x >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!((w >> (x & 31)) & 1u)) atomicOr(&gmap[x >> 5], 1u << (x & 31));
		} else {
			atomicOr(&s_map[x >> 5], 1u << (x & 31));
		}
	};
	auto clear_map = [&]() {
		if constexpr (GM) {
			uint4 *m4 = reinterpret_cast<uint4 *>(gmap);
			for (int k = tid; k < bm_words / 4; k += kM4Threads) m4[k] = make_uint4(0, 0, 0, 0);
		} else {
			uint4_alias *m4 = reinterpret_cast<uint4_alias *>(s_map);
			for (int k = tid; k < bm_words / 4; k += kM4Threads) m4[k] = make_uint4(0, 0, 0, 0);
		}
	};
	auto flag_set = [&]() { return *(volatile int *)&s_flag != 0; };
	// every wavefront reads the flag between two barriers, so that none of them can still be reading it when a later
	// phase sets it again (a wavefront delayed between the barrier and its read would otherwise take another branch)
	auto flag_snapshot = [&]() {
		__syncthreads();
		const int v = s_flag;
		__syncthreads();
		return v;
	};
	// the padded lists hold copies of a list's last entry and the lanes past a round's end re-read its last group:
	// both repeat real entries, which neither a bit test nor a mark minds.  Every wavefront may request cap / 16 entries.
	auto walk_test = [&](const uint4 *list, int list_n, const int32_t *xp, bool have_first, uint4 first, int64_t limit) {
		bool f = false, capped = false;
		int resume = 0;
		const unsigned long long e2 = seg_walk<PGQ_MEET4_DEPTH, false>(
		    list, list_n, wib, PGQ_MEET4_WAVES, xp, win, have_first, first, (unsigned long long)limit / PGQ_MEET4_WAVES, capped, resume,
		    [&](const int4 &v, bool, u32) { f |= (bit((u32)v.x) | bit((u32)v.y) | bit((u32)v.z) | bit((u32)v.w)) != 0; },
		    [&]() {
			    if (__any(f)) s_flag = 1;
			    return flag_set();
		    });
		entries += e2;
		if (__any(f)) s_flag = 1;
		if (capped) s_capped = 1;
	};
	auto walk_mark = [&](const uint4 *list, int list_n, const int32_t *xp, uint4 first) {
		bool capped = false;
		int resume = 0;
		const unsigned long long e2 = seg_walk<PGQ_MEET4_MARK_DEPTH, false>(
		    list, list_n, wib, PGQ_MEET4_WAVES, xp, win, true, first, (unsigned long long)cap / PGQ_MEET4_WAVES, capped, resume,
		    [&](const int4 &v, bool, u32) {
			    mark((u32)v.x);
			    mark((u32)v.y);
			    mark((u32)v.z);
			    mark((u32)v.w);
		    },
		    []() { return false; });
		entries += e2;
		if (capped) s_capped = 1;
	};
	// option meet_trace: per-workgroup {start, end, rows, longest row} in 10-ns ticks of the constant clock
	const unsigned long long t_begin = TRACE ? wall_clock64() : 0ull;
	unsigned long long t_row_max = 0, n_rows_done = 0;
	// Rows are handed out in queue order by an atomic counter (the first gridDim.x positions are the workgroups' own): the
	// long rows sit at the front, so they start first and a workgroup stuck in one simply takes fewer of the short ones —
	// with the static stride of round 3 a slot could draw two long rows, or a long one last.  Thread 0 draws the NEXT
	// position while the current row is processed (the atomic's round trip is off the critical path).
	// the sampled decision of the distinct sources rides here when the chain runs on the route memo's word (meet_prepass,
	// decide_mode 2): the LAST workgroup — its first row is one of the short ones, and the rows are handed out dynamically —
	// takes the sample in the bit map's LDS before its first row clears it
	if (!GM && sm.h_go && blockIdx.x == gridDim.x - 1 && bm_words >= kSampleSlots && !db->ball.go)
		sample_distinct_sources(sm.n, sm.src, sm.V, sm.meet_bytes, sm.edge_bytes, sm.out, sm.h_go, s_map);
	u32 job = blockIdx.x;
	for (;;) {
		__syncthreads(); // the previous row's flags, map and s_job are no longer read
		if (job >= n) break;
		u32 next_job = 0;
		if (tid == 0) next_job = gridDim.x + atomicAdd(&db->next_job, 1u);
		const unsigned long long t_row = TRACE ? wall_clock64() : 0ull;
		// the row's queue entry: everything the stage before knew about it, in three 16-byte loads (the queue is a few
		// tens of KB written moments ago), wave-uniform -> scalar registers
		const uint4 *ep = reinterpret_cast<const uint4 *>(qin.ent + queue_pos(qin, job, nf));
		const uint4 e0 = ep[0], e1 = ep[1], e2w = ep[2];
		const u32 workS = (u32)__builtin_amdgcn_readfirstlane((int)e2w.x), workD = (u32)__builtin_amdgcn_readfirstlane((int)e2w.y);
		const u32 row = (u32)__builtin_amdgcn_readfirstlane((int)e0.x), flags = (u32)__builtin_amdgcn_readfirstlane((int)e0.y);
		const u32 es = (u32)__builtin_amdgcn_readfirstlane((int)e0.z), ed = (u32)__builtin_amdgcn_readfirstlane((int)e0.w);
		const int so = __builtin_amdgcn_readfirstlane((int)e1.x), degS = __builtin_amdgcn_readfirstlane((int)e1.y);
		const int di = __builtin_amdgcn_readfirstlane((int)e1.z), degD = __builtin_amdgcn_readfirstlane((int)e1.w);
		const bool known4 = (flags & kEntKnown4) != 0, known3 = (flags & kEntKnown3) != 0;
		clear_map();
		if (tid == 0) {
			s_flag = 0;
			s_capped = 0;
		}
		__syncthreads();
		if (wib == 0) vertices += (u32)(degS + degD); // both one-hop lists as descriptors
		bool walk_fwd = workS <= workD; // which endpoint's two-hop neighbourhood is walked / marked: the smaller one
		int64_t result = kMeetOpen;   // wave-uniform
		int resume3 = 0;
		bool do3 = false, do4 = true;
		if (known3) {
			// k_meet3 excluded distances 1 and 2 and walked part of one endpoint's two-hop neighbourhood against the
			// other endpoint's one-hop list: the same test with 16 wavefronts, from the round it stopped in
			// k_meet3 walks from the endpoint with the longer two-hop walk when the other endpoint's list does not fit its
			// register set (512 ids); the bit map takes any list, so the side is chosen again by the walks' sizes — R-MAT-22:
			// a row that had followed k_meet3 into a hub's two-hop neighbourhood walked its full million-entry allowance,
			// 270 us for one row — and the walk is taken up where k_meet3 stopped only if the side is the same
			walk_fwd = workS <= workD;
			resume3 = walk_fwd == ((flags & kEntFwd) != 0) ? (int)(flags >> kEntResumeShift) : 0;
			const int32_t *set_list = walk_fwd ? radj + di : adj + so;
			const int set_n = walk_fwd ? degD : degS;
			for (int p = tid; p < set_n; p += kM4Threads) mark((u32)set_list[p]);
			if (wib == 0) entries += (unsigned long long)set_n;
			__syncthreads();
			do3 = true;
		} else if (!known4) {
			bool hit = false;
			for (int p = tid; p < degS; p += kM4Threads) hit |= (u32)adj[so + p] == ed;
			if (__any(hit) && lane == 0) s_flag = 1;
			if (flag_snapshot()) { // dst in N_out(src)
				result = 1;
				do4 = false;
			} else {
				const int64_t work_f = (int64_t)workS, work_b = (int64_t)workD; // the two-hop walks' sizes came with the row
				walk_fwd = work_f <= work_b;
				{ // the set: one-hop list of the endpoint that is not walked
					const int32_t *set_list = walk_fwd ? radj + di : adj + so;
					const int set_n = walk_fwd ? degD : degS;
					for (int p = tid; p < set_n; p += kM4Threads) mark((u32)set_list[p]);
				}
				__syncthreads();
				const int32_t *wl = walk_fwd ? adj + so : radj + di;
				const int wn = walk_fwd ? degS : degD;
				if (wib == 0) entries += (unsigned long long)(degS + degD);
				{ // distance 2: a common neighbour
					bool f = false;
					for (int p = tid; p < wn; p += kM4Threads) f |= bit((u32)wl[p]) != 0;
					if (__any(f) && lane == 0) s_flag = 1;
				}
				if (flag_snapshot()) {
					result = 2;
					do4 = false;
				} else if (min(work_f, work_b) > cap) {
					do4 = false; // stays open
				} else {
					do3 = true;
					if (max(work_f, work_b) > cap) do4 = false; // if the distance-3 walk finds nothing the row stays open
				}
			}
		}
		if (do3) {
			// distance 3: the cheaper two-hop walk against the other endpoint's one-hop set
			const uint4 *wl = (walk_fwd ? fdesc + so : rdesc + di) + resume3;
			const int wn = (walk_fwd ? degS : degD) - resume3;
			walk_test(wl, wn, walk_fwd ? padj : rpadj, false, make_uint4(0, 0, 0, 0), cap);
			if (flag_snapshot()) {
				result = 3;
				do4 = false;
			} else {
				if (s_capped) do4 = false;
				if (do4) clear_map(); // every wavefront is past its reads of the map (barriers above)
				__syncthreads();
			}
		}
		if (do4) {
			// distance 4: two-hop set of the walked endpoint, two-hop walk of the other one.  Both walks' first
			// descriptors are requested together (one round trip instead of two)
			const uint4 *mark_list = walk_fwd ? fdesc + so : rdesc + di, *test_list = walk_fwd ? rdesc + di : fdesc + so;
			const int mark_n = walk_fwd ? degS : degD, test_n = walk_fwd ? degD : degS;
			uint4 d_mark = make_uint4(0, 0, 0, 0), d_test = make_uint4(0, 0, 0, 0);
			if (lane < mark_n) d_mark = mark_list[lane]; // every wavefront holds the round's 64 descriptors (it takes every 16th request)
			if (lane < test_n) d_test = test_list[lane];
			// Stage A: the first 2048 entries of BOTH two-hop neighbourhoods at once.  A pair at distance 4 — nearly every row
			// that gets here — has ~16 common vertices among two such prefixes on the SF100-shaped graph (degree-biased
			// samples collide at sum of p_v^2 = E[deg^2] / (V E[deg]^2) = 3.8e-6 per pair of entries), so marking all 8.5 K
			// entries of the smaller neighbourhood before looking (two dependent rounds of requests, then the probe: three
			// round trips) is rarely needed: the lower half of the wavefronts requests and marks one group each of the marking
			// side, the upper half requests one group each of the other side in the SAME round trip and tests it after the
			// barrier.  A hit is a witness like any other (the marks are real entries of one neighbourhood, the tested entries
			// real entries of the other); no hit: the full procedure below, which marks the prefix again.
			int f4 = 0;
			constexpr int H = PGQ_MEET4_WAVES / 2;
			constexpr int kOneShot = 1 << 20; // request stride that leaves a wavefront exactly one request of the round (no refill)
			{
				int4 held = make_int4(0, 0, 0, 0);
				bool have = false;
				bool capped = false;
				int resume = 0;
				if (wib < H) {
					const unsigned long long e2 = seg_walk<1, false>(
					    mark_list, min(mark_n, 64), wib, kOneShot, walk_fwd ? padj : rpadj, win, true, d_mark, 0ull, capped, resume,
					    [&](const int4 &v, bool, u32) {
						    mark((u32)v.x);
						    mark((u32)v.y);
						    mark((u32)v.z);
						    mark((u32)v.w);
					    },
					    []() { return true; });
					entries += e2;
				} else {
					const unsigned long long e2 = seg_walk<1, false>(
					    test_list, min(test_n, 64), wib - H, kOneShot, walk_fwd ? rpadj : padj, win, true, d_test, 0ull, capped, resume,
					    [&](const int4 &v, bool, u32) {
						    held = v;
						    have = true;
					    },
					    []() { return true; });
					entries += e2;
				}
				__syncthreads();
				if (have && (bit((u32)held.x) | bit((u32)held.y) | bit((u32)held.z) | bit((u32)held.w))) s_flag = 1;
				f4 = flag_snapshot();
			}
			if (!f4) {
				walk_mark(mark_list, mark_n, walk_fwd ? padj : rpadj, d_mark);
				__syncthreads();
			}
			if (!f4 && !s_capped) {
				// Probe first: nearly every row that gets here IS at distance 4, and then about one entry in thirty of the other
				// endpoint's two-hop neighbourhood is marked — the first request (256 entries) holds a witness.  The cooperative
				// walk below opens with 32 requests (16 wavefronts x 2 in flight) before anyone looks at the flag: ~45 KB and
				// their bit tests per row, for an answer the first KB gives.  Two wavefronts take one request each (the two
				// behind the ones stage A tested).
				if (wib < 2) {
					bool f = false, capped = false;
					int resume = 0;
					const unsigned long long e2 = seg_walk<1, false>(
					    test_list, min(test_n, 64), H + wib, kOneShot, walk_fwd ? rpadj : padj, win, true, d_test, 0ull, capped, resume,
					    [&](const int4 &v, bool, u32) { f |= (bit((u32)v.x) | bit((u32)v.y) | bit((u32)v.z) | bit((u32)v.w)) != 0; },
					    []() { return true; });
					entries += e2;
					if (__any(f)) s_flag = 1;
				}
				f4 = flag_snapshot();
				if (!f4) {
					// no witness in the first 512 entries: the pair is far apart (or unreachable), or the marked set is tiny.
					// The other endpoint's two-hop neighbourhood may be the huge one (the smaller one was marked): it gets
					// `test_cap` entries, not the million of a marking walk — what is still open then is k_bibfs's kind of row
					// (R-MAT-22: such rows walked their full allowance, 270 us each, only to be handed on)
					walk_test(test_list, test_n, walk_fwd ? rpadj : padj, true, d_test, test_cap);
					f4 = flag_snapshot();
				}
			}
			if (f4) result = 4;
		}
		if (tid == 0) {
			out_rows[row] = result;
			if (result == kMeetOpen) {
				const MeetEntry e = { row, 0u, es, ed, (u32)so, (u32)degS, (u32)di, (u32)degD, workS, workD, 0u, 0u };
				queue_push(qout, row, e);
			}
			s_job = next_job;
		}
		if constexpr (TRACE) {
			t_row_max = max(t_row_max, wall_clock64() - t_row);
			n_rows_done++;
		}
		__syncthreads();
		job = s_job;
	}
	if (TRACE && tid == 0) {
		trace[4 * blockIdx.x] = t_begin;
		trace[4 * blockIdx.x + 1] = wall_clock64();
		trace[4 * blockIdx.x + 2] = n_rows_done;
		trace[4 * blockIdx.x + 3] = t_row_max;
	}
	__shared__ unsigned long long s_stat[2];
	__syncthreads();
	if (tid < 2) s_stat[tid] = 0;
	__syncthreads();
	if (lane == 0 && entries) atomicAdd(&s_stat[0], entries);
	if (tid == 0 && vertices) atomicAdd(&s_stat[1], (unsigned long long)vertices);
	__syncthreads();
	if (tid == 0) meet_add_stats(&db->m, 1, s_stat[0], s_stat[1]);
	meet_finalize(db, fin);
}

// ---- any distance, few rows: bidirectional BFS per row (k_bibfs) ---------------------------------------------------
// What is still open after the kernels above is far apart (distance >= 5), unreachable, or over their caps.  On a graph
// whose levels are expensive (R-MAT-22: ~1 ms per bottom-up level whatever the number of lanes) a handful of such rows
// would drag the whole lane-batched search along, although each of them has a small side: an unreachable pair is
// unreachable because one endpoint's closure is small, and a far pair's two frontiers stay far below the graph's
// size until they meet.  One 1024-thread workgroup per row runs the textbook bidirectional search, level-synchronous:
//     F = {src}, B = {dst} (visited bit maps, one per side; frontier vertex lists in global scratch), a = b = 0
//     expand the side whose frontier has fewer adjacency entries by one level; a newly reached vertex that the other
//     side has visited ends the search with distance a + b + 1
// Exactness: before an expansion F and B are disjoint, so the distance exceeds a + b; a vertex x reached at forward
// level a + 1 that B holds has backward level b exactly (a smaller one would put its forward predecessor into B as well,
// and the search would have ended a round earlier), so the first meeting gives the BFS distance.  An empty new frontier
// means that side's closure is complete: NULL, like the exhausted search of iterativelength.cpp:133-139.  A frontier over
// `cap` entries or `qcap` vertices leaves the row open.
template <bool GM>
__global__ __launch_bounds__(1024) void k_bibfs(MeetQueue qin, u32 max_rows,
                                                const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                                int64_t *__restrict__ out_rows, int64_t cap,
                                                int bm_words, int qcap, MeetDevBlock *__restrict__ db,
                                                u32 *__restrict__ gmaps, u32 *__restrict__ queues, MeetQueue qout,
                                                MeetHostBlock *__restrict__ fin) {
	const int64_t *const src = qin.src, *const dst = qin.dst;
	const u32 *const didx = qin.idx;
	MeetCounters *const mc = &db->m;
	extern __shared__ u32 s_map[]; // !GM: both maps, (bm_words + 4) words each; the spare words take the masked lanes' bits
	// rows still open, counted on the device (no host round trip before this launch); more than a handful: not this
	// kernel's job, the lane-batched search takes them
	const int64_t n = (int64_t)*qin.count;
	if (n > (int64_t)max_rows) {
		// not this kernel's job: the rows pass through to the next queue unchanged (the host reads one region whatever ran)
		for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
			qout.src[i] = src[i];
			qout.dst[i] = dst[i];
			qout.idx[i] = didx[i];
			reinterpret_cast<uint4 *>(qout.ent + i)[0] = reinterpret_cast<const uint4 *>(qin.ent + i)[0];
			reinterpret_cast<uint4 *>(qout.ent + i)[1] = reinterpret_cast<const uint4 *>(qin.ent + i)[1];
		}
		if (blockIdx.x == 0 && threadIdx.x == 0) *qout.count = (u32)n;
		meet_finalize(db, fin);
		return;
	}
	const int mw = bm_words + 4;
	u32 *const gmap = GM ? gmaps + (size_t)blockIdx.x * 2 * mw : nullptr;
	u32 *const qbase = queues + (size_t)blockIdx.x * 5 * qcap; // [side][parity][qcap], then the touched-word list [qcap]
	// GM: the two maps are 2 x V / 8 bytes per row (R-MAT-22: 1 MB) and a search that ends after a few hundred vertices used to
	// pay their full clear — 22 GB of stores per 2048 x 1024 cross product, most of the kernel's 7.7 ms.  The thread whose
	// atomicOr finds a word still zero lists it; the next row clears the listed words (a list over `qcap`: the full clear).
	u32 *const tlist = qbase + (size_t)4 * qcap;
	__shared__ u32 s_tn;
	__shared__ int s_found;
	__shared__ u32 s_cnt;
	__shared__ unsigned long long s_work;
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	const u32 dummy = (u32)bm_words * 32u; // bit 0 of the first spare word
	unsigned long long entries = 0;
	u32 vertices = 0;
	auto or_rtn = [&](int side, u32 x) -> u32 {
		if constexpr (GM) { // look first: a bit that is set needs no read-modify-write (hubs are reached over and over)
			// (tried: no look for levels of a few thousand entries — one dependent round trip less per level; configs[1]'s k_bibfs
			// stayed at 107-110 us: a level there is ~10 us of five dependent DRAM trips and four barriers whatever its size)
			const u32 w = __hip_atomic_load(&gmap[side * mw + (x >> 5)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((w >> (x & 31)) & 1u) return w;
			return atomicOr(&gmap[side * mw + (x >> 5)], 1u << (x & 31));
		} else {
			return atomicOr(&s_map[side * mw + (x >> 5)], 1u << (x & 31));
		}
	};
	auto word_of = [&](int side, u32 x) -> u32 {
		if constexpr (GM) return __hip_atomic_load(&gmap[side * mw + (x >> 5)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else return s_map[side * mw + (x >> 5)];
	};
	if (tid == 0) s_tn = ~0u; // the maps arrive with whatever the last launch left: the first row clears them all
	for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
		__syncthreads();
		const int64_t s = src[i], d = dst[i]; // open rows: ids in range, src != dst, both have edges
		const u32 row = didx[i] & ~kMeetKnown4Bit;
		if constexpr (GM) {
			const u32 tn = s_tn;
			if (tn > (u32)qcap) {
				uint4 *m4 = reinterpret_cast<uint4 *>(gmap); // mw is a multiple of 4, slices are 16-byte aligned
				for (int k = tid; k < 2 * mw / 4; k += 1024) m4[k] = make_uint4(0, 0, 0, 0);
			} else {
				for (u32 k = tid; k < tn; k += 1024) gmap[tlist[k]] = 0;
			}
			__syncthreads();
			if (tid == 0) s_tn = 0;
		} else {
			for (int k = tid; k < 2 * mw; k += 1024) s_map[k] = 0;
		}
		if (tid == 0) s_found = 0;
		__syncthreads();
		if (tid == 0) {
			(void)or_rtn(0, (u32)s);
			(void)or_rtn(1, (u32)d);
			qbase[0] = (u32)s;
			qbase[2 * qcap] = (u32)d;
			if constexpr (GM) {
				tlist[0] = (u32)s >> 5;
				tlist[1] = (u32)mw + ((u32)d >> 5);
				s_tn = 2;
			}
		}
		int nf[2] = { 1, 1 }, lvl[2] = { 0, 0 }, par[2] = { 0, 0 };
		int64_t work[2] = { off[s + 1] - off[s], roff[d + 1] - roff[d] };
		int64_t result = kMeetOpen;
#ifdef PGQ_MEET3_ROWTRACE
		const unsigned long long bt0 = wall_clock64();
		unsigned long long bt_work = 0;
#endif
		__syncthreads();
		for (;;) {
			const int side = work[0] <= work[1] ? 0 : 1; // 0: forward from src, 1: backward from dst
			if (work[side] > cap) break;
#ifdef PGQ_MEET3_ROWTRACE
			bt_work += (unsigned long long)work[side];
#endif
			const int64_t *xoff = side ? roff : off;
			const int32_t *xadj = side ? radj : adj;
			const u32 *cur = qbase + (size_t)(side * 2 + par[side]) * qcap;
			u32 *nxt = qbase + (size_t)(side * 2 + (par[side] ^ 1)) * qcap;
			if (tid == 0) {
				s_cnt = 0;
				s_work = 0;
				vertices += (u32)nf[side];
			}
			__syncthreads();
			bool hit = false;
			const unsigned long long e2 = meet_walk_chunks<PGQ_BIBFS_DEPTH>(
			    reinterpret_cast<const int32_t *>(cur), nf[side], wib, 16, xoff, xadj,
			    [&](const int4 &v, u32 valid, u32) {
				    u32 xs[4] = { (u32)v.x, (u32)v.y, (u32)v.z, (u32)v.w };
				    u32 old[4], oth[4];
#pragma unroll
				    for (int k = 0; k < 4; k++) xs[k] = ((valid >> k) & 1u) ? xs[k] : dummy; // no per-lane branches around memory operations
#pragma unroll
				    for (int k = 0; k < 4; k++) old[k] = or_rtn(side, xs[k]);
#pragma unroll
				    for (int k = 0; k < 4; k++) oth[k] = word_of(side ^ 1, xs[k]);
				    if constexpr (GM) { // words this search has just made non-zero (the masked lanes' spare word included)
#pragma unroll
					    for (int k = 0; k < 4; k++) {
						    const bool first = old[k] == 0u;
						    const u64 m = __ballot(first);
						    if (m) {
							    u32 base = 0;
							    if (lane == 0) base = atomicAdd(&s_tn, (u32)__popcll(m));
							    base = (u32)__shfl((int)base, 0);
							    const u32 slot = base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
							    if (first && slot < (u32)qcap) tlist[slot] = (u32)(side * mw) + (xs[k] >> 5);
						    }
					    }
				    }
#pragma unroll
				    for (int k = 0; k < 4; k++) {
					    const bool in = (valid >> k) & 1u;
					    const bool fresh = in && !((old[k] >> (xs[k] & 31)) & 1u);
					    hit |= in && ((oth[k] >> (xs[k] & 31)) & 1u);
					    const u64 m = __ballot(fresh);
					    if (m) {
						    u32 base = 0;
						    if (lane == 0) base = atomicAdd(&s_cnt, (u32)__popcll(m));
						    base = (u32)__shfl((int)base, 0);
						    const u32 slot = base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
						    if (fresh && slot < (u32)qcap) nxt[slot] = xs[k];
					    }
				    }
			    },
			    [&]() {
				    if (__any(hit)) s_found = 1;
				    return *(volatile int *)&s_found != 0;
			    });
			if (lane == 0) entries += e2;
			if (__any(hit)) s_found = 1;
			__syncthreads();
			if (s_found) {
				result = lvl[0] + lvl[1] + 1;
				break;
			}
			const u32 nn = s_cnt;
			if (nn == 0) { // this side's closure is complete and never met the other one
				result = -1;
				break;
			}
			if (nn > (u32)qcap) break;
			unsigned long long w = 0;
			for (u32 p = tid; p < nn; p += 1024) {
				const u32 v = nxt[p];
				w += (unsigned long long)(xoff[v + 1] - xoff[v]);
			}
			for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
			if (lane == 0 && w) atomicAdd(&s_work, w);
			__syncthreads();
			work[side] = (int64_t)s_work;
			nf[side] = (int)nn;
			lvl[side]++;
			par[side] ^= 1;
			__syncthreads(); // s_cnt / s_work are reset by the next round
		}
#ifdef PGQ_MEET3_ROWTRACE
		if (tid == 0 && wall_clock64() - bt0 > 3000)
			printf("bibfs row %u: %.1f us, levels %d + %d, frontier entries expanded %llu, last frontiers %d / %d, result %lld\n", row, (wall_clock64() - bt0) * 0.01,
			       lvl[0], lvl[1], bt_work, nf[0], nf[1], (long long)result);
#endif
		if (tid == 0) {
			out_rows[row] = result;
			if (result == kMeetOpen) {
				const MeetEntry e = { row, (didx[i] & kMeetKnown4Bit) ? kEntKnown4 : 0u, (u32)s, (u32)d, 0u, 0u, 0u, 0u };
				queue_push(qout, row, e);
			}
		}
	}
	__shared__ unsigned long long s_stat[2];
	__syncthreads();
	if (tid < 2) s_stat[tid] = 0;
	__syncthreads();
	for (int o = 32; o > 0; o >>= 1) entries += __shfl_xor(entries, o);
	if (lane == 0 && entries) atomicAdd(&s_stat[0], entries);
	if (tid == 0 && vertices) atomicAdd(&s_stat[1], (unsigned long long)vertices);
	__syncthreads();
	if (tid == 0) meet_add_stats(mc, 2, s_stat[0], s_stat[1]);
	meet_finalize(db, fin);
}

// ---- path emission -------------------------------------------------------------------------------------------------
// [src, e1, v1, ..., ek, dst] for the rows the pre-pass answered (shortest_path.cpp:149-204): the edge of a hop is the
// FIRST slot of the parent holding the child (shortest_path.cpp:23-30).  One wavefront per row.
__global__ void k_path_counts(int64_t n, const int64_t *__restrict__ len, int64_t *__restrict__ cnt) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) cnt[i] = len[i] >= 0 ? 2 * len[i] + 1 : 0; // open (kMeetOpen, kMeetOpen4) and NULL rows: nothing here
}
__global__ __launch_bounds__(256) void k_emit_paths(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                    const int64_t *__restrict__ len, const MeetPath *__restrict__ rec,
                                                    const int64_t *__restrict__ poff, const int64_t *__restrict__ off,
                                                    const int32_t *__restrict__ adj, const int64_t *__restrict__ edge_ids,
                                                    int64_t *__restrict__ child, int64_t child_cap, int64_t *__restrict__ out_off) {
	// One 256-thread workgroup per row, wavefront h = hop h of the path (at most four hops here): the four first-slot
	// searches are independent once the inner vertices are known, so they run side by side, each 256 entries per round
	// trip (round 3: one wavefront per row, the hops one after the other, 64 entries per trip — 28 us for 4096 rows).
	const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
	const int64_t i = (int64_t)blockIdx.x; // 64-bit row index
	if (i >= n) return;
	const int64_t k = len[i];
	if (k < 0 || h >= (k == 0 ? 1 : k)) return; // open / NULL rows: nothing here; wavefronts past the last hop
	const int64_t base_out = poff[i];
	if (base_out + 2 * k + 1 > child_cap) return; // the caller's buffer is too small: it is told so (lengths stay valid)
	int64_t *out = child + base_out;
	if (h == 0 && lane == 0) {
		out_off[i] = base_out;
		out[0] = src[i];
	}
	if (k == 0) return;
	const MeetPath r = rec[i];
	// vertices of the path: src, v1 .. v(k-1), dst
	auto vertex = [&](int j) -> int64_t { return j == 0 ? src[i] : (j == (int)k ? dst[i] : (j == 1 ? r.v1 : (j == 2 ? r.v2 : r.v3))); };
	const int64_t p = vertex(h), c = vertex(h + 1);
	const int64_t b = off[p], e = off[p + 1];
	int64_t slot = -1;
	for (int64_t base = b; base < e && slot < 0; base += 256) {
		int32_t x[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int64_t t = base + 64 * u + lane;
			x[u] = t < e ? adj[t] : -1;
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const u64 m = __ballot((int64_t)x[u] == c);
			if (m && slot < 0) slot = base + 64 * u + (__ffsll((long long)m) - 1);
		}
	}
	if (lane == 0) {
		out[2 * h + 1] = slot < 0 ? -1 : (edge_ids ? edge_ids[slot] : slot);
		out[2 * h + 2] = c;
	}
}

// ---- how many distinct sources? ---------------------------------------------------------------------------------------
// The pre-pass costs one two-hop walk per ROW, the lane-batched search one lane per distinct SOURCE: for cross-product
// shaped inputs (few sources x many destinations, the binder's shape) the latter wins by orders of magnitude.  A
// strided sample of the rows goes through an LDS hash set, thread 0 inverts E[distinct] = U (1 - (1 - 1/U)^sample) and
// compares the two cost estimates ON THE DEVICE: the pre-pass kernels are launched straight behind and return at once
// when the flag says no, so the host waits once per call instead of once for the decision and once for the result.
// Ends a chain that holds the source-centric kernels alone (meet_prepass, ball_mode 3): reports like the last stage kernel would.
__global__ void k_chain_end(MeetDevBlock *__restrict__ db, MeetHostBlock *__restrict__ fin) { meet_finalize(db, fin); }

// ball_go (nullable): the source-centric kernel in front of this one has taken the call: no sample, no pre-pass.
__global__ __launch_bounds__(1024) void k_meet_decide(int64_t n, const int64_t *__restrict__ src, int64_t V, double meet_bytes,
                                                     double edge_bytes, MeetDecision *__restrict__ out, u32 *__restrict__ h_go,
                                                     const u32 *__restrict__ ball_go) {
	__shared__ u32 s_set[kSampleSlots];
	if (ball_go && *ball_go) {
		if (threadIdx.x == 0) out->go = 0;
		return;
	}
	sample_distinct_sources(n, src, V, meet_bytes, edge_bytes, out, h_go, s_set);
}

__global__ void k_apply_open(int64_t nd, const u32 *__restrict__ didx, const int64_t *__restrict__ dlen,
                             int64_t *__restrict__ out) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < nd) out[didx[j] & ~kMeetKnown4Bit] = dlen[j];
}

// The two queue regions of a workspace (A: offset 0, B: offset n) and the device / host statistic blocks.
static int meet_buffers(Workspace *ws, int64_t n, MeetQueue q[2], MeetDevBlock **db, MeetHostBlock **hb) {
	hipStream_t st = ws->stream;
	const bool fresh = ws->meet_cnt.cap < sizeof(MeetDevBlock);
	PGQ_TRY(ws->meet_cnt.reserve(sizeof(MeetDevBlock)));
	*db = ws->meet_cnt.as<MeetDevBlock>();
	// the chain's last kernel leaves the block zeroed; a fresh block, or one a failed call left behind, is cleared here
	if (fresh || !ws->meet_cnt_clean) PGQ_HIP_TRY(hipMemsetAsync(*db, 0, sizeof(MeetDevBlock), st));
	ws->meet_cnt_clean = false;
	const size_t rows = (size_t)std::max<int64_t>(n, 1) * 2;
	PGQ_TRY(ws->def_src.reserve(rows * 8));
	PGQ_TRY(ws->def_dst.reserve(rows * 8));
	PGQ_TRY(ws->def_idx.reserve(rows * 4));
	PGQ_TRY(ws->def_ent.reserve(rows * sizeof(MeetEntry)));
	for (int k = 0; k < 2; k++) {
		q[k].src = ws->def_src.as<int64_t>() + (size_t)k * n;
		q[k].dst = ws->def_dst.as<int64_t>() + (size_t)k * n;
		q[k].idx = ws->def_idx.as<u32>() + (size_t)k * n;
		q[k].ent = ws->def_ent.as<MeetEntry>() + (size_t)k * n;
		q[k].count = nullptr;
		q[k].count_back = nullptr;
		q[k].cap = (u32)n;
	}
	*hb = static_cast<MeetHostBlock *>(ws->h_meet);
	(*hb)->done = 0;
	return PGQ_OK;
}
// waits for the chain and takes over what its last workgroup wrote into the pinned block
static int meet_wait(Workspace *ws, MeetHostBlock *hb, bool report_is_last = true) {
	// meet_spin_wait (off as shipped): poll the report the chain's last workgroup writes into pinned memory instead of asking
	// the runtime for the stream — 5-6 us per call (2048 rows 55.6 -> 50.5 us, 8192 rows 78.7 -> 73.0, one row 26 -> 21).  The
	// call then returns on the kernels' own system-scope fences (every workgroup fences before its ticket, the last one before
	// the report) without the end-of-kernel release a stream synchronisation adds; results are complete by construction, but a
	// consumer on ANOTHER stream is no longer covered by the runtime's guarantee — hence an option, not the default.
	// (only where the report IS the chain's last word: shortestpath's scan and list emission run behind the stage kernels —
	// tests/soak_gpu.py caught a call that returned on the report with its lists still unwritten)
	if (options().meet_spin_wait && report_is_last && !options().profile) {
		const auto t0 = std::chrono::steady_clock::now();
		for (u32 it = 0;; it++) {
			if (*(volatile u32 *)&hb->done == 1u) {
				std::atomic_thread_fence(std::memory_order_acquire);
				ws->meet_cnt_clean = true;
				return PGQ_OK;
			}
			__builtin_ia32_pause();
			if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
		}
	}
	PGQ_TRY(wait_stream(ws->stream, &ws->ev_block));
	KernelTimer::flush();
	if (hb->done != 1) return fail(PGQ_ERR_HIP, "the pre-pass chain did not report back (statistics block not written)");
	ws->meet_cnt_clean = true;
	return PGQ_OK;
}
static void meet_attributes() {
	static std::atomic<int> attr_set { 0 };
	if (attr_set.load()) return;
	(void)hipFuncSetAttribute((const void *)k_meet4d<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void)hipFuncSetAttribute((const void *)k_meet4d<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void)hipFuncSetAttribute((const void *)k_meet4<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void)hipFuncSetAttribute((const void *)k_bibfs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	attr_set.store(1);
}

// debugging aid (option meet_trace): where k_src_ball's time goes
static int print_ball_trace(const unsigned long long *b_trace, u32 nseg, u32 open) {
	unsigned long long t[17];
	PGQ_HIP_TRY(hipMemcpy(t, b_trace, sizeof(t), hipMemcpyDeviceToHost));
	const double seg = (double)std::max<u32>(nseg, 1) * 100.0;
	fprintf(stderr, "[pgq] k_src_ball trace: longest single phase, us: rows+clear %.1f, S1 %.1f, S2 %.1f, in-ball tests %.1f, in-list scans %.1f, distance 4 %.1f, "
	        "output %.1f, between segments %.1f\n", t[9] * 0.01, t[10] * 0.01, t[11] * 0.01, t[12] * 0.01, t[13] * 0.01, t[14] * 0.01, t[15] * 0.01, t[16] * 0.01);
	fprintf(stderr, "[pgq] k_src_ball trace: %u segments, %u rows left open, us per segment: rows+clear %.1f, S1 %.1f, S2 %.1f, in-ball tests %.1f, in-list scans %.1f, "
	        "distance 4 %.1f, output %.1f, between segments %.1f; longest segment %.1f us\n",
	        nseg, open, t[0] / seg, t[1] / seg, t[2] / seg, t[3] / seg, t[4] / seg, t[5] / seg, t[6] / seg, t[7] / seg, (double)t[8] * 0.01);
	return PGQ_OK;
}

// Runs the pre-pass over n rows (device memory, or pinned host memory the device can address: the chunk entry points
// hand their staging block over as it is); rows it answers get their hop count (or -1 for NULL) in d_out, the others end
// up in ws->open_src / open_dst / open_idx and are counted in *n_open.  The whole chain — decision (large inputs),
// k_meet3, the bit-map kernel, the bidirectional search for a handful of leftovers — is launched back to back; every
// kernel reads what it needs (the go flag, the number of rows still open) from device memory and appends what it leaves
// open to the next one's queue, the last one reports into the pinned block: the host launches 2-4 kernels and waits ONCE.
// decide: k_meet_decide compares `meet_bytes` with the lanes' cost for the sampled number of distinct sources
// (lanes_cost_bytes) first; *ran = false when it said no (nothing was written to d_out).
// ball_mode (round 6): 1 = k_ball_segments + k_src_ball open the chain and the device decides from the number of source
// runs whether the source-centric kernel takes the call (then every stage kernel behind it returns at once and *ball_ran
// = true: the open rows are what IT left); 2 = it always does (tests); 0 = the chain starts with the stage kernels.
int meet_prepass(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, int64_t *d_out,
                 u32 *n_open, MeetPathsOut *po, int decide_mode, double meet_bytes, double edge_bytes, bool *ran, int *observed_go,
                 int ball_mode, bool *ball_ran, double *est_sources) {
	const bool paths = po != nullptr;
	if (ball_ran) *ball_ran = false;
	if (est_sources) *est_sources = -1.0; // the decision kernel's estimate of the distinct sources, when it ran
	if (paths || !c->rseg || !c->fdesc || !c->rdesc || n < 2) ball_mode = 0;
	// ball_mode 3: the route memo says the source-centric kernel took these buffers last time — the chain is its two kernels
	// and a one-thread report, without the stage kernels that would only return at once behind it (65,536 one-wavefront
	// workgroups of k_meet3 and 256 of k_meet4d starting up to read one word: 28 us of a 0.36-ms call).  If it declines this
	// time, *ball_ran stays false, *ran = false, and the caller runs the chain again without it.
	const bool ball_only = ball_mode == 3;
	if (ball_only) ball_mode = 1;
	// decide_mode 1: k_meet_decide in front of the chain, its flag gates every stage kernel on the device.  2: the route
	// memo says the last call on these buffers was answered here: the chain runs ungated and the sample rides in k_meet4d's
	// launch (its last workgroup, in the bit map's LDS) — *observed_go gets its verdict for the memo, -1 when none was taken
	// (12 us of kernel + a launch gap in front of every 65,536-row call otherwise; tried first: the sample on a second
	// stream beside the chain — the extra launch and wait on the host cost what the kernel did)
	if (observed_go) *observed_go = -1;
	{ // the ride needs k_meet4d (distance-only flow) with its bit map in LDS and large enough to lend: else the gate again
		const Options &o = options();
		const int bmw = (int)((c->V + 127) / 128) * 4;
		const size_t budget = (size_t)std::min(150, std::max(0, o.meet4_lds_kb)) * 1024;
		if (decide_mode == 2 && !(o.meet4 && !paths && (size_t)bmw * 4 + 2048 <= budget && bmw >= kSampleSlots)) decide_mode = 1;
	}
	const bool decide = decide_mode == 1;
	u32 *h_go = reinterpret_cast<u32 *>(static_cast<char *>(ws->h_meet) + 4104);
	SampleArgs ride { meet_bytes, edge_bytes, nullptr, nullptr, n, d_src, c->V };
	if (decide_mode == 2) {
		PGQ_TRY(ws->route_dec.reserve(sizeof(MeetDecision)));
		*h_go = 0;
		ride.out = ws->route_dec.as<MeetDecision>();
		ride.h_go = h_go; // taken inside k_meet4d's launch when its map is in LDS and large enough; else no verdict this call
	}
	hipStream_t st = ws->stream;
	pgq_stats_t &S = tstats().s;
	const Options &opt = options();
	if (ran) *ran = true;
	MeetQueue q[2];
	MeetDevBlock *db = nullptr;
	MeetHostBlock *hb = nullptr;
	PGQ_TRY(meet_buffers(ws, n, q, &db, &hb));
	if (paths) PGQ_TRY(ws->meet_rec.reserve((size_t)n * sizeof(MeetPath)));
	MeetPath *rec = paths ? ws->meet_rec.as<MeetPath>() : nullptr;
	const u32 *d_go = decide ? &db->dec.go : nullptr;
	// what is left after k_meet3 (distance >= 4, or over its caps): the bit-map kernels, launched straight behind on a
	// fixed grid — they read the row count from the device.  The vertex bit map sits in LDS when it fits (V <= ~1.2 M);
	// above that every workgroup gets a slice of a global buffer (L2-resident: 0.5 MB at V = 4 M).
	const int bm_words = (int)((c->V + 127) / 128) * 4;
	const size_t lds_budget = (size_t)std::min(150, std::max(0, opt.meet4_lds_kb)) * 1024;
	const bool lds_map = (size_t)bm_words * 4 + 2048 <= lds_budget;
	const size_t gm_budget = (size_t)std::max(0, opt.meet4_global_mb) << 20;
	const bool run4 = opt.meet4 && (lds_map || (size_t)bm_words * 4 <= gm_budget);
	// k_bibfs serves the few rows the two-hop kernels leave open (far apart, unreachable, over the caps).  Launching it
	// costs ~12 us of stream time even when no row is open, so it stays in the chain only while this CSR has shown such
	// rows: the first call runs it; a call that ends with every row answered before it switches it off, and any later
	// call that leaves rows open (they go to the lane-batched search, same answers) switches it on again.
	const bool run_bi = !paths && opt.bibfs_rows > 0 && c->meet_far_rows.load(std::memory_order_relaxed) != 0;
	const int mwb = bm_words + 4;
	const bool bi_lds = (size_t)2 * mwb * 4 + 2048 <= lds_budget; // k_bibfs: both sides' maps in LDS when they fit
	const int qcap = std::max(1024, opt.bibfs_queue);
	// round 6: how many rows it takes and on how many workgroups follows the graph and what the call before left: on a
	// graph whose levels are expensive (R-MAT-22: a lane batch for the 15,800 far / unreachable rows of a 2 M-row cross product
	// costs 19 ms) a bidirectional search per row on every CU is far cheaper than whole-graph levels for a few thousand rows,
	// while a graph that has never shown more than a handful keeps the 64-workgroup launch (12 us when nothing is open)
	const int far_rows = c->meet_far_rows.load(std::memory_order_relaxed);
	const int bibfs_rows = opt.bibfs_rows <= 0 ? 0 : std::max(opt.bibfs_rows, (int)std::min<int64_t>(opt.bibfs_rows_max, c->E / 4096));
	// (sized by what the last call left AND by this call's rows: a handle's first call knows nothing of the former, and 64
	// workgroups for the 35,000 far rows of an R-MAT-22 cross product made that call 11 ms longer than the ones after it)
	// (on graphs whose maps are global — V past the LDS: there far rows are the rule; a graph with LDS maps keeps the small
	// grid until a call has left far rows: every workgroup owns 2.6 MB of queues, 512 of them 1.3 GB allocated on first use)
	const int want_grid = std::max(far_rows / 16, bi_lds ? 0 : (int)std::min<int64_t>(n / 64, 1 << 20));
	const u32 bi_grid = (u32)std::min(std::max(std::max(1, opt.bibfs_grid), std::min(want_grid, 2 * device_cus())), std::max(1, bibfs_rows));
	// k_meet4d hands rows out dynamically: a grid of exactly the workgroups the chip holds (meet4_grid_mult = 2 per CU).
	// A row alone on its CU is through in ~15 us, beside a second one in ~20 (the phases of a row are short bursts of
	// instructions from 16 wavefronts, and two workgroups share the CU's issue slots): small calls, whose ~2 % of open rows
	// do not fill 256 CUs anyway, get one workgroup per CU (8192 rows: 0.086 -> 0.076 ms, 2048 rows: 0.062 -> 0.053 ms)
	const bool small_call = !paths && n <= (int64_t)opt.meet_small_rows;
	u32 grid4 = (u32)std::min<int64_t>(n, (int64_t)device_cus() * std::max(1, paths ? 4 : (small_call ? 1 : opt.meet4_grid_mult)));
	// round 6: a chunk-sized call leaves a few dozen rows open (2 % of 2048): a workgroup per CU for them starts 256 x 1024
	// threads that find nothing to do and, worse, fills every CU — the chunk calls of DuckDB's other worker threads (one per
	// DataChunk per thread, iterativelength.cpp:34) queue behind it instead of running beside it (tools/chunk_mt.cpp: 8 threads
	// reached 47 M rows/s, 1.7 x one thread).  One workgroup per 16 rows, at least 16: rows are handed out dynamically anyway.
	if (small_call) grid4 = std::min<u32>(grid4, (u32)std::max<int64_t>(16, n / 16));
	size_t maps_bytes = 0;
	if (run4 && !lds_map) {
		grid4 = (u32)std::max<size_t>(1, std::min<size_t>((size_t)std::min<int64_t>(n, device_cus()), gm_budget / ((size_t)bm_words * 4)));
		maps_bytes = (size_t)grid4 * bm_words * 4;
	}
	const size_t bi_map_words = (run_bi && !bi_lds) ? (size_t)bi_grid * 2 * mwb : 0;
	const size_t bi_bytes = run_bi ? (bi_map_words + (size_t)bi_grid * 5 * qcap) * 4 + 64 : 0;
	if (maps_bytes + bi_bytes > 0) PGQ_TRY(ws->meet_maps.reserve(maps_bytes + bi_bytes + 64));
	u32 *gmaps = ws->meet_maps.as<u32>();
	u32 *bi_maps = gmaps ? gmaps + (maps_bytes + 15) / 16 * 4 : nullptr;
	meet_attributes();
	// stage k appends to region k & 1 and counts in db->count[k]
	for (int k = 0; k < 2; k++) q[k].count = &db->count[k];
	if (run4 && !paths) q[0].count_back = &db->count[3]; // k_meet3 -> k_meet4d: long rows from the front, the others from the back
	unsigned long long *d_trace = nullptr;
	if (opt.meet_trace && run4 && !paths) {
		PGQ_TRY(ws->meet_trace.reserve((size_t)grid4 * 32));
		d_trace = ws->meet_trace.as<unsigned long long>();
		PGQ_HIP_TRY(hipMemsetAsync(d_trace, 0, (size_t)grid4 * 32, st));
	}
	const int last_stage = run_bi ? 2 : (run4 ? 1 : 0);
	// shortestpath on a large input: if the decision kernel calls the pre-pass off, nothing writes d_out — the list layout
	// below must then see "no list" everywhere (-1 in every row), not what the buffer happened to hold
	if (decide && paths) PGQ_HIP_TRY(hipMemsetAsync(d_out, 0xFF, (size_t)n * 8, st));
	// ---- round 6: the source-centric kernels open the chain (pgq_ball.h) ----
	unsigned long long *b_trace = nullptr;
	if (ball_mode) {
		// its vertex bit map: LDS when one workgroup's map + 23 KB of row state fit (two workgroups per CU when both do), else a
		// slice of the global buffer the bit-map kernels use (they run only when this one declines)
		const size_t row_state = 23 * 1024, lds_all = 160 * 1024;
		const bool ball_lds = (size_t)bm_words * 4 + row_state <= std::min(lds_all, lds_budget + row_state);
		unsigned grid_b = 0;
		size_t ball_maps = 0;
		if (ball_lds) {
			grid_b = (unsigned)device_cus() * ((PGQ_BALL_WAVES >= 8 && 2 * ((size_t)bm_words * 4 + row_state) <= lds_all) ? 2u : 1u);
		} else if ((size_t)bm_words * 4 <= gm_budget) {
			grid_b = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)device_cus() * 2, gm_budget / ((size_t)bm_words * 4)));
			ball_maps = (size_t)grid_b * bm_words * 4;
		}
		if (opt.ball_grid > 0) grid_b = std::min(grid_b, (unsigned)opt.ball_grid);
		if (n <= (int64_t)opt.meet_small_rows) grid_b = std::min<unsigned>(grid_b, (unsigned)std::max<int64_t>(8, n / 16)); // (as for k_meet4d below: room for the other threads' chunks)
		if (grid_b == 0) {
			ball_mode = 0; // no room for the maps: the older routes
		} else {
			if (ball_maps > ws->meet_maps.cap) {
				PGQ_TRY(ws->meet_maps.reserve(std::max(ball_maps, maps_bytes + bi_bytes + 64)));
				gmaps = ws->meet_maps.as<u32>();
				bi_maps = gmaps + (maps_bytes + 15) / 16 * 4;
			}
			PGQ_TRY(ws->ball_segs.reserve((size_t)n * 4));
			static std::atomic<int> ball_attr { 0 };
			if (!ball_attr.load()) {
				(void)hipFuncSetAttribute((const void *)k_src_ball<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 137 * 1024);
				(void)hipFuncSetAttribute((const void *)k_src_ball<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 137 * 1024);
				ball_attr.store(1);
			}
			BallRule rule;
			const double mean_deg = (double)c->E / (double)std::max<int64_t>(c->V, 1);
			rule.seg_floor = 1024.0 * (double)std::max(0, opt.ball_seg_kb);
			rule.seg_bytes = 4.0 * c->two_hop_mean + 32.0 * mean_deg + 64.0;
			rule.row_bytes = 4.0 * std::min(mean_deg, 64.0) + (c->rhead ? 0.0 : 128.0) + 32.0; // the first 64 entries of the in-list, the line its position sits in (without rhead), the row
			// a ball in a global map: every mark is a look in DRAM and an atomic there (R-MAT-22: ~5 ns per entry and workgroup
			// against ~1 ns in LDS) ...
			if (!ball_lds) rule.seg_bytes *= 8.0;
			// ... and what this kernel leaves open is the pre-pass's to answer, at that kernel's measured bytes per row: the share of
			// rows the last call on this graph shape left open (0 before the first) prices it.  R-MAT-22, 2048 x 1024: 1.7 % open x
			// 1.7 MB per row = 28 KB per row — the lane batches (12 ms) are then the cheaper route, not this one (16.7 ms).
			{
				const double bpr = meet_bytes > 0 && n > 0 ? meet_bytes / (double)n : c->meet_bpr.load(std::memory_order_relaxed);
				rule.row_bytes += c->ball_open_frac.load(std::memory_order_relaxed) * std::max(0.0, bpr);
			}
			rule.meet_bytes = meet_bytes;
			rule.edge_bytes = edge_bytes > 0 ? edge_bytes : (double)c->E;
			rule.bias = opt.ball_bias;
			rule.mode = ball_mode;
			rule.V = c->V;
			const int64_t nwin = (n + kBallRows - 1) / kBallRows;
			u32 seg_rows = (u32)kBallRows; // chunk-sized calls: shorter segments, more workgroups per source (k_ball_segments)
			if (n <= (int64_t)opt.meet_small_rows) {
				seg_rows = 64;
				while ((int)seg_rows * 2 <= std::min(kBallRows, std::max(64, opt.ball_seg_rows_small))) seg_rows *= 2;
			}
			KernelTimer kt(st, K_BALL);
			hipLaunchKernelGGL(k_ball_segments, dim3((unsigned)std::min<int64_t>(nwin, (int64_t)device_cus())), dim3(kBallRows), 0, st, n, d_src,
			                   ws->ball_segs.as<u32>(), db, seg_rows);
			const int64_t capb = std::max(1, opt.ball_cap), tcap = std::max(1, opt.ball_test_cap);
			if (opt.meet_trace) {
				PGQ_TRY(ws->ball_trace.reserve(256));
				b_trace = ws->ball_trace.as<unsigned long long>();
				PGQ_HIP_TRY(hipMemsetAsync(b_trace, 0, 256, st));
			}
#define PGQ_BALL(G, T, LDS)                                                                                                  \
	hipLaunchKernelGGL((k_src_ball<G, T>), dim3(grid_b), dim3(kBallRows), LDS, st, n, d_src, d_dst, c->V, c->off, c->roff, c->fdesc, \
	                   c->rdesc, c->padj, c->rpadj, c->rseg, opt.ball_head_mb > 0 ? c->rhead : (const uint4 *)nullptr, ws->ball_segs.as<u32>(), d_out, capb, tcap, \
	                   bm_words, db, gmaps, q[0], rule, b_trace, seg_rows)
			if (ball_lds && b_trace) PGQ_BALL(false, true, (size_t)bm_words * 4);
			else if (ball_lds) PGQ_BALL(false, false, (size_t)bm_words * 4);
			else if (b_trace) PGQ_BALL(true, true, 0);
			else PGQ_BALL(true, false, 0);
#undef PGQ_BALL
			kt.stop();
		}
	}
	if (ball_only && ball_mode) {
		hipLaunchKernelGGL(k_chain_end, dim3(1), dim3(64), 0, st, db, hb);
		PGQ_TRY(meet_wait(ws, hb));
		const MeetHostBlock &h = *hb;
		if (h.bad) return fail(PGQ_ERR_INVALID_ARG, "src/dst rowid out of range [0,V)");
		if (!h.ball_go) { // not this time: nothing was answered
			S.algo_bytes[K_BALL] += 16.0 * (double)n;
			if (ran) *ran = false;
			*n_open = (u32)n;
			return PGQ_OK;
		}
		if (ball_ran) *ball_ran = true;
		if (b_trace) PGQ_TRY(print_ball_trace(b_trace, h.ball_nseg, h.ball_open));
		ws->open_src = q[0].src;
		ws->open_dst = q[0].dst;
		ws->open_idx = q[0].idx;
		S.meet_pairs += n - (int64_t)h.ball_open;
		S.edges_scanned += (int64_t)h.ball_entries;
		S.algo_bytes[K_BALL] += 4.0 * (double)h.ball_entries + 16.0 * (double)h.ball_descs + 32.0 * (double)n + 24.0 * (double)h.ball_nseg;
		S.ball_segments += h.ball_nseg;
		S.ball_calls++;
		if (est_sources) *est_sources = (double)h.ball_nrun; // (exact: the source runs it counted)
		*n_open = h.ball_open;
		return PGQ_OK;
	}
	// (tried in round 4: the decision kernel on a stream of its own beside k_meet3, which polls a stop flag — the event
	// record / wait pair costs what the 12 us kernel does, and the polled word must be spread over many lines)
	if (decide)
		hipLaunchKernelGGL(k_meet_decide, dim3(1), dim3(1024), 0, st, n, d_src, c->V, meet_bytes, edge_bytes, &db->dec, (u32 *)nullptr,
		                   (const u32 *)&db->ball.go);
	{
		// calls too small to fill the chip are bound by the longest row, not by bandwidth: more requests in flight shorten
		// every row (meet_cap_small is a cap of their own; 4096 .. 16384 measured within 3 % of each other: it ships equal
		// to meet_cap)
		const bool small = small_call;
		const int64_t cap = std::max(1, paths ? opt.meet_cap_paths : (small ? opt.meet_cap_small : opt.meet_cap));
		const bool bigv = c->V > (1 << 20);
		KernelTimer kt(st, K_MEET);
		const unsigned resident = (unsigned)device_cus() * 32 / kMeetWPB; // more workgroups than the chip holds at once: up to 8 rounds
		const dim3 grid((unsigned)std::min<int64_t>((n + kMeetWPB - 1) / kMeetWPB, (int64_t)std::max(1, opt.meet_grid_mult) * resident));
		MeetHostBlock *fin = last_stage == 0 ? hb : nullptr;
#define PGQ_MEET3(P, B, D)                                                                                               \
	hipLaunchKernelGGL((k_meet3<P, B, D>), grid, dim3(64 * kMeetWPB), 0, st, n, d_src, d_dst, c->V, c->off, c->adj, c->roff,  \
	                   c->radj, c->fdesc, c->rdesc, c->padj, c->rpadj, c->fwork, c->rwork, d_out, rec, cap, d_go, db, q[0], fin)
		if (paths) {
			if (bigv) PGQ_MEET3(true, true, PGQ_MEET3_DEPTH);
			else PGQ_MEET3(true, false, PGQ_MEET3_DEPTH);
		} else if (small && n <= (int64_t)opt.meet_wide_rows && (opt.meet_wide_rows_always || (double)c->E * 4.0 > 256e6)) {
			// chunk-sized calls on a graph whose adjacency does not fit the Infinity Cache (a list request is a DRAM round trip,
			// ~3.5 us under load): several wavefronts per row (k_meet3w: 97 VGPRs, four wavefronts per SIMD = 4096 on the chip) —
			// four while all rows are resident at once, else two.  Measured: R-MAT-22 x 1024 pairs 49 -> 41 us; the SF100-shaped
			// graph (160 MB of padded lists, cache resident) 24.1 -> 25.5 us at 1024 rows, 29.1 -> 29.7 at 2048: not taken there
#define PGQ_MEET3W(B, W)                                                                                                  \
	hipLaunchKernelGGL((k_meet3w<B, W>), dim3((unsigned)n), dim3(64 * W), 0, st, n, d_src, d_dst, c->V, c->off, c->adj, c->roff, c->radj, \
	                   c->fdesc, c->rdesc, c->padj, c->rpadj, c->fwork, c->rwork, d_out, cap, d_go, db, q[0], fin)
			const bool four = n * 4 <= (int64_t)device_cus() * 16;
			if (bigv && four) PGQ_MEET3W(true, 4);
			else if (bigv) PGQ_MEET3W(true, 2);
			else if (four) PGQ_MEET3W(false, 4);
			else PGQ_MEET3W(false, 2);
#undef PGQ_MEET3W
		} else if (small) {
			if (bigv) PGQ_MEET3(false, true, PGQ_MEET3_DEPTH_SMALL);
			else PGQ_MEET3(false, false, PGQ_MEET3_DEPTH_SMALL);
		} else {
			if (bigv) PGQ_MEET3(false, true, PGQ_MEET3_DEPTH);
			else PGQ_MEET3(false, false, PGQ_MEET3_DEPTH);
		}
#undef PGQ_MEET3
		kt.stop();
	}
	int open_stage = 0; // the stage whose queue holds what is open at the end
	if (run4) {
		const size_t lds = lds_map ? (size_t)bm_words * 4 : 0;
		const int64_t cap4 = (int64_t)std::max(1, opt.meet4_cap);
		MeetHostBlock *fin = last_stage == 1 ? hb : nullptr;
		{
			KernelTimer kt(st, K_MEET4);
#define PGQ_MEET4(G)                                                                                                     \
	hipLaunchKernelGGL((k_meet4<true, G>), dim3(grid4), dim3(1024), lds, st, q[0], c->V, c->off, c->adj, c->roff, c->radj,     \
	                   c->fdesc, c->rdesc, c->padj, c->rpadj, d_out, rec, cap4, bm_words, db, gmaps, q[1], fin)
#define PGQ_MEET4D(G, T)                                                                                                    \
	hipLaunchKernelGGL((k_meet4d<G, T>), dim3(grid4), dim3(kM4Threads), lds, st, q[0], c->adj, c->radj, c->fdesc, c->rdesc, c->padj, \
	                   c->rpadj, d_out, cap4, (int64_t)std::max(1, opt.meet4_test_cap), bm_words, db, gmaps, q[1], fin, d_trace, ride)
			if (paths && lds_map) PGQ_MEET4(false);
			else if (paths) PGQ_MEET4(true);
			else if (lds_map && d_trace) PGQ_MEET4D(false, true);
			else if (lds_map) PGQ_MEET4D(false, false);
			else if (d_trace) PGQ_MEET4D(true, true);
			else PGQ_MEET4D(true, false);
#undef PGQ_MEET4D
#undef PGQ_MEET4
			kt.stop();
		}
		open_stage = 1;
	}
	// a handful of rows still open (far apart, unreachable, over the caps): one bidirectional search each, so that the
	// lane-batched search — whole-graph levels — only starts for what really needs it.  The kernel checks the count itself.
	if (run_bi) {
		u32 *queues = bi_maps + bi_map_words;
		const int64_t capb = (int64_t)std::max(1, opt.bibfs_cap);
		const MeetQueue &qi = q[open_stage];
		MeetQueue qo2 = q[open_stage ^ 1]; // the other region: the stage that filled it has been read by now
		qo2.count = &db->count[2];
		qo2.count_back = nullptr; // one-ended: what k_bibfs leaves open is counted in count[2] alone (q[0] carries k_meet3's two-ended counter)
		{
			KernelTimer kt(st, K_BIBFS);
			if (bi_lds)
				hipLaunchKernelGGL(k_bibfs<false>, dim3(bi_grid), dim3(1024), (size_t)2 * mwb * 4, st, qi, (u32)bibfs_rows,
				                   c->off, c->adj, c->roff, c->radj, d_out, capb, bm_words, qcap, db, bi_maps, queues, qo2, hb);
			else
				hipLaunchKernelGGL(k_bibfs<true>, dim3(bi_grid), dim3(1024), 0, st, qi, (u32)bibfs_rows, c->off, c->adj,
				                   c->roff, c->radj, d_out, capb, bm_words, qcap, db, bi_maps, queues, qo2, hb);
			kt.stop();
		}
		ws->open_src = qo2.src;
		ws->open_dst = qo2.dst;
		ws->open_idx = qo2.idx;
		open_stage = 2;
	} else {
		ws->open_src = q[open_stage].src;
		ws->open_dst = q[open_stage].dst;
		ws->open_idx = q[open_stage].idx;
	}
	// shortestpath: the lists of the rows answered so far are laid out (element counts -> exclusive scan) and written in the
	// same chain — the total comes back in the pinned block beside the statistics, so the call still waits once (round 3:
	// three waits — chain, scan total, emission)
	int64_t *h_total = reinterpret_cast<int64_t *>(static_cast<char *>(ws->h_meet) + 4096);
	if (paths) {
		PGQ_TRY(ws->meet_poff.reserve((size_t)(n + 1) * 8 * 2));
		int64_t *cnt = ws->meet_poff.as<int64_t>() + (n + 1), *poff = ws->meet_poff.as<int64_t>();
		hipLaunchKernelGGL(k_path_counts, dim3(blocks_for(n)), dim3(256), 0, st, n, d_out, cnt);
		PGQ_HIP_TRY(hipMemsetAsync(cnt + n, 0, 8, st));
		size_t tmp = 0;
		PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, cnt, poff, (int)(n + 1), st));
		PGQ_TRY(ws->scan_tmp.reserve(tmp + 16));
		PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws->scan_tmp.p, tmp, cnt, poff, (int)(n + 1), st));
		PGQ_HIP_TRY(hipMemcpyAsync(h_total, poff + n, 8, hipMemcpyDeviceToHost, st));
		KernelTimer kt(st, K_RECON);
		hipLaunchKernelGGL(k_emit_paths, dim3((unsigned)n), dim3(256), 0, st, n, d_src, d_dst, d_out, rec, poff, c->off, c->adj,
		                   c->edge_ids, po->d_child, po->child_cap, po->d_out_off);
		kt.stop();
	}
	PGQ_TRY(meet_wait(ws, hb, !paths));
	if (decide_mode == 2 && observed_go && *h_go) *observed_go = (int)*h_go - 1;
	if (paths) po->total = *h_total;
	const MeetHostBlock &h = *hb;
	if (d_trace) { // debugging aid: where k_meet4d's time goes
		std::vector<unsigned long long> t((size_t)grid4 * 4);
		PGQ_HIP_TRY(hipMemcpy(t.data(), d_trace, t.size() * 8, hipMemcpyDeviceToHost));
		unsigned long long t0 = ~0ull, t1 = 0, rows = 0, longest = 0, first_end = ~0ull, last_start = 0, most = 0;
		for (u32 b = 0; b < grid4; b++) {
			if (!t[4 * b + 1]) continue;
			t0 = std::min(t0, t[4 * b]);
			t1 = std::max(t1, t[4 * b + 1]);
			first_end = std::min(first_end, t[4 * b + 1]);
			last_start = std::max(last_start, t[4 * b]);
			rows += t[4 * b + 2];
			most = std::max(most, t[4 * b + 2]);
			longest = std::max(longest, t[4 * b + 3]);
		}
		fprintf(stderr, "[pgq] k_meet4d trace: %u workgroups, %llu rows (%u long first), at most %llu per workgroup, span %.1f us, last start +%.1f us, "
		        "first end +%.1f us, longest row %.1f us\n",
		        grid4, rows, h.count[0], most, (double)(t1 - t0) * 0.01, (double)(last_start - t0) * 0.01,
		        (double)(first_end - t0) * 0.01, (double)longest * 0.01);
	}
	if (b_trace && h.ball_go) PGQ_TRY(print_ball_trace(b_trace, h.ball_nseg, h.ball_open));
	if (ball_mode && h.ball_go) { // the source-centric kernel took the call: what is open sits in region 0, counted by itself
		if (h.bad) return fail(PGQ_ERR_INVALID_ARG, "src/dst rowid out of range [0,V)");
		if (ball_ran) *ball_ran = true;
		ws->open_src = q[0].src;
		ws->open_dst = q[0].dst;
		ws->open_idx = q[0].idx;
		S.meet_pairs += n - (int64_t)h.ball_open;
		S.edges_scanned += (int64_t)h.ball_entries;
		// 4 B per adjacency entry (the balls' lists, the destinations' in-lists, the distance-4 walks), 16 B per slot descriptor,
		// per row its ids, its destination's list position and its result (32 B), per segment its position and offsets (24 B)
		S.algo_bytes[K_BALL] += 4.0 * (double)h.ball_entries + 16.0 * (double)h.ball_descs + 32.0 * (double)n + 24.0 * (double)h.ball_nseg;
		S.ball_segments += h.ball_nseg;
		S.ball_calls++;
		if (est_sources) *est_sources = (double)h.ball_nrun; // (exact: the source runs it counted)
		*n_open = h.ball_open;
		return PGQ_OK;
	}
	if (ball_mode) // it looked at the rows (8 B per row, twice) and declined
		S.algo_bytes[K_BALL] += 16.0 * (double)n;
	if (decide && est_sources) *est_sources = h.dec.estimate;
	if (decide && !h.dec.go) {
		if (ran) *ran = false;
		*n_open = (u32)n;
		return PGQ_OK;
	}
	if (h.bad) return fail(PGQ_ERR_INVALID_ARG, "src/dst rowid out of range [0,V)");
	const u32 open = h.count[open_stage];
	if (!paths && opt.bibfs_rows > 0) {
		const u32 before_bi = h.count[run4 ? 1 : 0]; // rows open when k_bibfs was (or would have been) launched
		c->meet_far_rows.store((int)std::min<u32>(before_bi, 1u << 30), std::memory_order_relaxed); // their number sizes the next call's grid
	}
	S.meet_pairs += n - (int64_t)open;
	S.edges_scanned += (int64_t)(h.entries[0] + h.entries[1] + h.entries[2]);
	// 4 B per adjacency entry / one-hop id, 16 B per slot descriptor; k_meet3: per row its ids (16 B), the four offsets and
	// two walk sizes of its endpoints (40 B) and its result (8 B); the bit-map kernel: per queued row its entry (48 B)
	// and its result (8 B)
	S.algo_bytes[K_MEET] += 4.0 * (double)h.entries[0] + 16.0 * (double)h.vertices[0] + 64.0 * (double)n;
	S.algo_bytes[K_MEET4] += 4.0 * (double)h.entries[1] + 16.0 * (double)h.vertices[1] + 56.0 * (double)(h.count[0] + h.count_back);
	S.algo_bytes[K_BIBFS] += 4.0 * (double)h.entries[2] + 16.0 * (double)h.vertices[2];
	*n_open = open;
	return PGQ_OK;
}

// shortestpath: the caller reserved less than 9 elements per row (paths_reserve_mb) and the lists did not fit: the list
// offsets (meet_poff) and inner vertices (meet_rec) of the chain that just ran are still in the workspace — written again
int meet_reemit_paths(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, const int64_t *d_out, MeetPathsOut *po) {
	hipStream_t st = ws->stream;
	PGQ_HIP_TRY(hipMemsetAsync(po->d_out_off, 0, (size_t)n * 8, st));
	KernelTimer kt(st, K_RECON);
	hipLaunchKernelGGL(k_emit_paths, dim3((unsigned)n), dim3(256), 0, st, n, d_src, d_dst, d_out, ws->meet_rec.as<MeetPath>(), ws->meet_poff.as<int64_t>(),
	                   c->off, c->adj, c->edge_ids, po->d_child, po->child_cap, po->d_out_off);
	kt.stop();
	return PGQ_OK;
}

// The sampled decision alone, waited for: shortestpath on a large input asks before it reserves 72 bytes per row for the
// lists of the rows the pre-pass would answer (a cross product is called off: nothing of it would be used).
int meet_decide_alone(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, double meet_bytes, double edge_bytes, bool *go) {
	PGQ_TRY(ws->route_dec.reserve(sizeof(MeetDecision)));
	u32 *h_go = reinterpret_cast<u32 *>(static_cast<char *>(ws->h_meet) + 4104);
	*h_go = 0;
	hipLaunchKernelGGL(k_meet_decide, dim3(1), dim3(1024), 0, ws->stream, n, d_src, c->V, meet_bytes, edge_bytes,
	                   ws->route_dec.as<MeetDecision>(), h_go, (const u32 *)nullptr);
	PGQ_HIP_TRY(hipStreamSynchronize(ws->stream));
	*go = *h_go == 2;
	return PGQ_OK;
}

// ---- iterativelengthbidirectional: every row through the bidirectional search ---------------------------------------------
// The reference's IterativeLengthBidirectionalFunction (iterativelength_bidirectional.cpp:43-153) is meant to search
// forward from src and backward from dst over the transposed CSR until the two meet (it is unreachable from the binder
// and indexes its backward CSR wrongly).  Here that is k_bibfs for EVERY row: k_bidir_classify answers
// what needs no search (NULL -> NULL, src == dst -> 0, an endpoint without edges in its direction -> NULL) and queues
// the rest; k_bibfs takes them one 1024-thread workgroup per row; rows over its caps stay open for the caller
// (the lane-batched search).  Same answers as iterativelength (the BFS distance), different schedule.
__global__ void k_bidir_classify(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst, int64_t V,
                                 const int64_t *__restrict__ off, const int64_t *__restrict__ roff, int64_t *__restrict__ out,
                                 MeetDevBlock *__restrict__ db, MeetQueue q) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t r = 0;
	bool open = false;
	int64_t s = 0, d = 0;
	if (i < n) {
		s = src[i], d = dst[i];
		r = kMeetOpen;
		if (s < 0) r = -1; // NULL row
		else if (s >= V || d < 0 || d >= V) {
			db->m.bad = 1;
			r = -1;
		} else if (s == d) r = 0;
		else if (off[s + 1] == off[s] || roff[d + 1] == roff[d]) r = -1; // no path can exist
		out[i] = r;
		open = r == kMeetOpen;
	}
	const u64 m = __ballot(open);
	if (!m) return;
	const int lane = threadIdx.x & 63;
	u32 base = 0;
	if (lane == 0) base = atomicAdd(q.count, (u32)__popcll(m));
	base = __shfl(base, 0);
	if (open) {
		const u32 p = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
		q.src[p] = s;
		q.dst[p] = d;
		q.idx[p] = (u32)i;
	}
}

int meet_bidirectional(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, int64_t *d_out,
                       u32 *n_open) {
	hipStream_t st = ws->stream;
	pgq_stats_t &S = tstats().s;
	const Options &opt = options();
	MeetQueue q[2];
	MeetDevBlock *db = nullptr;
	MeetHostBlock *hb = nullptr;
	PGQ_TRY(meet_buffers(ws, n, q, &db, &hb));
	for (int k = 0; k < 2; k++) q[k].count = &db->count[k];
	hipLaunchKernelGGL(k_bidir_classify, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, d_dst, c->V, c->off, c->roff, d_out, db, q[0]);
	const int bm_words = (int)((c->V + 127) / 128) * 4, mwb = bm_words + 4;
	const size_t lds_budget = (size_t)std::min(150, std::max(0, opt.meet4_lds_kb)) * 1024;
	const bool bi_lds = (size_t)2 * mwb * 4 + 2048 <= lds_budget;
	const int qcap = std::max(1024, opt.bibfs_queue);
	const u32 grid = (u32)std::min<int64_t>(n, device_cus());
	const size_t map_words = bi_lds ? 0 : (size_t)grid * 2 * mwb;
	PGQ_TRY(ws->meet_maps.reserve((map_words + (size_t)grid * 5 * qcap) * 4 + 64));
	u32 *maps = ws->meet_maps.as<u32>();
	u32 *queues = maps + map_words;
	meet_attributes();
	const int64_t capb = (int64_t)std::max(1, opt.bibfs_cap);
	{
		KernelTimer kt(st, K_BIBFS);
		if (bi_lds)
			hipLaunchKernelGGL(k_bibfs<false>, dim3(grid), dim3(1024), (size_t)2 * mwb * 4, st, q[0], 0xFFFFFFFFu, c->off, c->adj,
			                   c->roff, c->radj, d_out, capb, bm_words, qcap, db, maps, queues, q[1], hb);
		else
			hipLaunchKernelGGL(k_bibfs<true>, dim3(grid), dim3(1024), 0, st, q[0], 0xFFFFFFFFu, c->off, c->adj, c->roff, c->radj,
			                   d_out, capb, bm_words, qcap, db, maps, queues, q[1], hb);
		kt.stop();
	}
	ws->open_src = q[1].src;
	ws->open_dst = q[1].dst;
	ws->open_idx = q[1].idx;
	PGQ_TRY(meet_wait(ws, hb));
	const MeetHostBlock &h = *hb;
	if (h.bad) return fail(PGQ_ERR_INVALID_ARG, "src/dst rowid out of range [0,V)");
	S.meet_pairs += n - (int64_t)h.count[1];
	S.edges_scanned += (int64_t)h.entries[2];
	S.algo_bytes[K_BIBFS] += 4.0 * (double)h.entries[2] + 16.0 * (double)h.vertices[2];
	*n_open = h.count[1];
	return PGQ_OK;
}

// lengths and list offsets (shifted by `base`: their payload was appended there) of the rows the lane-batched search
// answered, scattered back to row order
__global__ void k_apply_open_paths(int64_t nd, const u32 *__restrict__ didx, const int64_t *__restrict__ dlen,
                                   const int64_t *__restrict__ doff, int64_t base, int64_t *__restrict__ out_len,
                                   int64_t *__restrict__ out_off) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nd) return;
	const u32 row = didx[j] & ~kMeetKnown4Bit;
	out_len[row] = dlen[j];
	if (dlen[j] >= 0) out_off[row] = base + doff[j];
}
int meet_apply_paths(Workspace *ws, int64_t nd, const int64_t *d_len, const int64_t *d_off, int64_t base,
                     int64_t *d_out_len, int64_t *d_out_off) {
	hipLaunchKernelGGL(k_apply_open_paths, dim3(blocks_for(nd)), dim3(256), 0, ws->stream, nd, ws->open_idx, d_len,
	                   d_off, base, d_out_len, d_out_off);
	return PGQ_OK;
}

int meet_apply(Workspace *ws, int64_t nd, const int64_t *d_len, int64_t *d_out) {
	hipLaunchKernelGGL(k_apply_open, dim3(blocks_for(nd)), dim3(256), 0, ws->stream, nd, ws->open_idx, d_len, d_out);
	return PGQ_OK;
}

} // namespace pgq

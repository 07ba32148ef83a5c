// pgq_cheapest.hip — cheapest_path_length on MI355X: batched, frontier-driven label-correcting SSSP.
//
// Replaces TemplatedBatchBellmanFord / TemplatedBellmanFord / CheapestPathLengthFunction
// (src/core/functions/scalar/cheapest_path_length.cpp:52-89, :91-136, :138-163).
//
// The reference sweeps every vertex x every out-edge x every lane until nothing changes, relaxing with
//     new = dist[v] + w;  if (new < dist[n]) dist[n] = new            (cheapest_path_length.cpp:29-36)
// For non-negative weights that relaxation is monotone (IEEE + is monotone in its left operand for a fixed
// non-negative right operand), so the fixpoint does not depend on the order of relaxations: the result is the
// minimum over paths of the left-to-right fold of the weights, bit for bit, for int64 and for double.  We
// therefore only relax out of vertices whose distance improved in the previous round (delta = "changed since
// last round" frontier), with atomicMin on the 64-bit pattern (non-negative doubles order like their bits).
// Negative weights are rejected (PGQ_ERR_UNSUPPORTED): the reference does not terminate on negative cycles
// and its max/2 sentinel leaks for unreachable lanes.
//
// Layout: dist[V][LC] (vertex-major, LC = 64 lanes -> one 512-byte row per vertex, fully coalesced per wave),
// dirty[parity][V] = 64-bit mask of lanes of v that improved in the previous round.  A lane is one distinct
// source.  Rounds are kernel launches: improvements made in round r are consumed in round r+1, so no
// intra-kernel visibility between XCDs is needed.
#include <hipcub/hipcub.hpp>

#include <cstddef>
#include <mutex>
#include <limits>
#include <type_traits>

#include "pgq_search.h"
#include <chrono>
#include <thread>
#include <vector>
#include <string>

namespace pgq {

static constexpr int LC = 64;

template <typename T> struct Inf;
template <> struct Inf<int64_t> {
	static constexpr int64_t bits = std::numeric_limits<int64_t>::max() / 2; // cheapest_path_length.cpp:15
};
template <> struct Inf<double> {
	// bit pattern of DBL_MAX / 2 = 0x7FDFFFFFFFFFFFFF
	static constexpr int64_t bits = 0x7FDFFFFFFFFFFFFFll;
};

// Label STORAGE of the batched relaxation (round 5): int64 weights whose largest possible path sum fits 31 bits
// (w_max x V < 2^31 - 1: weights 1..999 on the 448 K-vertex knows graph are at 4.5 x 10^8) keep their labels as int32 —
// a label row is 256 bytes instead of 512, and the rows are what the relaxation moves (2 rows per relaxed edge).  All
// arithmetic and every comparison stay in the 64-bit domain of the reference (cheapest_path_length.cpp:29-36): a stored
// label is widened on load, "unlabelled" (0x7FFFFFFF) to the reference's max / 2 sentinel, and a candidate is only
// narrowed when it is stored, where the precondition makes it exact.  Doubles and wider int64 ranges keep 8-byte labels.
template <typename DT> struct Lab;
template <> struct Lab<int64_t> {
	static constexpr int64_t unlabelled(int64_t inf_bits) { return inf_bits; }
	static __device__ __forceinline__ int64_t widen(int64_t v, int64_t) { return v; }
	static __device__ __forceinline__ int64_t min_into(int64_t *p, int64_t cand, int64_t) {
		return (int64_t)atomicMin((long long *)p, (long long)cand);
	}
};
template <> struct Lab<int32_t> {
	static constexpr int32_t kUnlabelled = 0x7FFFFFFF;
	static constexpr int32_t unlabelled(int64_t) { return kUnlabelled; }
	static __device__ __forceinline__ int64_t widen(int32_t v, int64_t inf_bits) { return v == kUnlabelled ? inf_bits : (int64_t)v; }
	static __device__ __forceinline__ int64_t min_into(int32_t *p, int64_t cand, int64_t inf_bits) {
		return widen(atomicMin(p, (int32_t)cand), inf_bits);
	}
};
__global__ void k_fill32(int32_t *__restrict__ p, int64_t n, int32_t value) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) p[i] = value;
}
__global__ void k_fill64(int64_t *__restrict__ p, int64_t n, int64_t value) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) p[i] = value;
}

// sources of the batch: dist[src][lane] = 0, dirty, queued
template <typename DT>
__global__ void k_cheapest_init(const int32_t *__restrict__ usrc, int64_t U, int64_t base, DT *__restrict__ dist,
                                u64 *__restrict__ dirty, u32 *__restrict__ tflag, u32 tepoch,
                                int32_t *__restrict__ touched, u32 *__restrict__ qcount, u32 *__restrict__ tcount,
                                int32_t *__restrict__ q) {
	int64_t g = base + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= U || g >= base + LC) return;
	const int v = usrc[g];
	const int l = (int)(g - base);
	dist[(size_t)v * LC + l] = 0;
	atomicOr(&dirty[v], 1ull << l);
	q[atomicAdd(qcount, 1u)] = v; // distinct sources -> distinct vertices
	tflag[v] = tepoch;
	touched[atomicAdd(tcount, 1u)] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// One round of the batched relaxation: one wavefront per changed vertex v, lane l = search l.  Every lane relaxes its
// own label over v's out-edges: coalesced 512-byte rows of dist[n][*].  Two things keep it from being a plain Jacobi
// sweep:
//  * ORDER (optional, relax_delta_div): only labels below `thr` are expanded this round, the others stay dirty.
//  * BOUND: a lane never expands a vertex whose label already reaches the largest tentative label among the lane's own
//    destinations (`bound`, refreshed every round): with non-negative weights nothing beyond can improve an answer.
// Neither changes the fixpoint at the destinations (every vertex on a cheaper path has a smaller label than the bound
// and is expanded after its last improvement), nor the left-to-right fold of the reference (cheapest_path_length.cpp:29-36),
// so int64 and double results stay bit-identical.  Labels are compared as their (non-negative) bit patterns.
//
// What a round costs is the number of DEPENDENT memory round trips a wavefront makes per vertex (the per-round trace of
// the weighted knows graph: ~420 K changed vertices with 4-6 useful edges each, for a dozen rounds), so the kernel is
// written to keep that chain short:
//  * the edges are handled eight at a time: adjacency+weights, then the eight label rows, then the eight atomicMin,
//    then — lanes 0..7, one improved neighbour each — the returning atomicOr on the neighbour's dirty word, whose old
//    value 0 says "first to dirty it this round: append it" (no separate queue flag), and the first-touch flag;
//  * appends to the next queue / the touched list are collected per wavefront in LDS and written 64 at a time (one
//    atomicAdd per VERTEX on the list counters serialised at ~12 ns each: 5 ms per round);
//  * the header of the wavefront's next vertex (dirty word, label row, list bounds) is requested before the current
//    vertex's edges are walked, the vertex after that is read from the queue.
// ---------------------------------------------------------------------------------------------------------------
// Compile-time variants of k_relax for tuning sweeps (tools/build_variants.sh, selected with PGQ_HIP_LIB); the defaults
// are what ships.
//   PGQ_RELAX_UNROLL   edges per dependent trip
//   PGQ_RELAX_WAVES    wavefronts per SIMD asked of the register allocator
//   PGQ_RELAX_NVFIRST  1: decide which of the trip's edges count before their label rows are requested (fewer rows, one
//                      more dependent step) — measured together with SEGCOND once: 117 -> 130 ms per 512 pairs
//   PGQ_RELAX_SEGCOND  1: a lane that cannot improve an edge's label reads v's first word instead of its own word of the
//                      row (tools/relax_model.py: 4.3 of a row's 8 segments hold a lane that still can)
#ifndef PGQ_RELAX_UNROLL
#define PGQ_RELAX_UNROLL 8
#endif
#ifndef PGQ_RELAX_WAVES
#define PGQ_RELAX_WAVES 6
#endif
#ifndef PGQ_RELAX_NVFIRST
#define PGQ_RELAX_NVFIRST 0
#endif
#ifndef PGQ_RELAX_SEGCOND
#define PGQ_RELAX_SEGCOND 0
#endif
static constexpr int kRelaxUnroll = PGQ_RELAX_UNROLL;
static constexpr int kPendCap = 64 + kRelaxUnroll;

// wave-level buffered append: lanes with `fresh` add n to the wavefront's pending list (LDS), 64 entries go out at once
__device__ __forceinline__ void relax_append(int *s_buf, int &pend, bool fresh, int n, int32_t *__restrict__ list,
                                             u32 *__restrict__ counter, int lane) {
	const u64 fm = __ballot(fresh);
	if (!fm) return;
	if (fresh) s_buf[pend + __popcll(fm & ((1ull << lane) - 1ull))] = n;
	pend += __popcll(fm);
	__builtin_amdgcn_wave_barrier();
	if (pend >= 64) {
		u32 base = 0;
		if (lane == 0) base = atomicAdd(counter, 64u);
		base = (u32)__builtin_amdgcn_readfirstlane((int)base);
		list[base + lane] = s_buf[lane];
		const int rest = pend - 64; // < kRelaxUnroll
		const int r = lane < rest ? s_buf[64 + lane] : 0;
		__builtin_amdgcn_wave_barrier();
		if (lane < rest) s_buf[lane] = r;
		pend = rest;
		__builtin_amdgcn_wave_barrier();
	}
}

template <typename T, typename DT>
__global__ __launch_bounds__(256, PGQ_RELAX_WAVES) void k_relax(const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                               const T *__restrict__ w, DT *__restrict__ dist,
                                               u64 *__restrict__ dirty_cur, u64 *__restrict__ dirty_nxt,
                                               const int32_t *__restrict__ qcur, const u32 *__restrict__ nq_ptr,
                                               int32_t *__restrict__ qnxt, u32 *__restrict__ nq_nxt,
                                               u32 *__restrict__ tflag, u32 tepoch, int32_t *__restrict__ touched,
                                               u32 *__restrict__ tcount, u64 *__restrict__ relaxed_edges, long long thr,
                                               const long long *__restrict__ bound, long long *__restrict__ min_deferred,
                                               u32 *__restrict__ relaxed_vertices, int sorted, T wcap,
                                               int heavy_pass, u64 *__restrict__ heavy_ctr, int32_t *__restrict__ hv,
                                               u64 *__restrict__ hmask, u32 *__restrict__ hstart, u32 *__restrict__ hmap,
                                               const DT *__restrict__ other, long long *__restrict__ mu,
                                               u32 *__restrict__ alive, const long long *__restrict__ thr_ptr) {
	// `other` != nullptr: one side of the bidirectional search (relax_batches_bidir).  Lane l is one (src, dst) PAIR, `dist` this
	// side's labels and `other` the other side's; `bound` is the pair's best known src -> dst length mu[l] (both sides prune
	// with it), `*thr_ptr` the distance cap both sides expand under.  A vertex expanded here that carries a label of the other
	// side closes a path: mu[l] = min(mu[l], label + other label).  alive[l] = 1: the lane left a labelled vertex unexpanded
	// (over the cap) this round — a side without one has exhausted its closure.
	// Long lists are not walked by the wavefront that finds them (a round would take as long as its longest list: one
	// hub of 1 800 edges = 230 dependent trips).  Pass 0 (heavy_pass = 0) appends such a vertex to the heavy list — one
	// packed 64-bit atomicAdd hands out its entry index and the first of its chunk numbers together — and writes
	// hmap[chunk] = entry; pass 1 (a second launch, items = chunks) walks kHeavyChunk edges per wavefront.  Over
	// weight-sorted lists only the prefix under the cap counts: a list is heavy if its 129th edge is still under it.
	constexpr int64_t kHeavyMin = 128;
	constexpr int64_t kHeavyChunk = 64;
	constexpr int UNR = kRelaxUnroll;
	__shared__ int s_pq[4][kPendCap];
	__shared__ int s_pt[4][kPendCap];
	// what the workgroup hands to the round's counters goes through LDS first: ~5 atomics per WAVEFRONT on one cache
	// line (two list tails, three statistics) were 25 K serialised L2 operations per launch — ~150 us, the whole
	// time of a round with few changed vertices
	__shared__ u64 s_edges;
	__shared__ u32 s_expanded;
	__shared__ long long s_min_def;
	__shared__ int s_pend[2][4];
	__shared__ u32 s_base[2];
	if (threadIdx.x == 0) {
		s_edges = 0;
		s_expanded = 0;
		s_min_def = 0x7FFFFFFFFFFFFFFFll;
	}
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int wib = threadIdx.x >> 6;
	const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const u32 nwaves = (gridDim.x * blockDim.x) >> 6;
	const u32 nq = heavy_pass ? (u32)(*heavy_ctr & 0xFFFFFFFFull) : *nq_ptr;
	long long my_bound = bound[lane];
	const long long mu_b = my_bound; // (bidirectional) the pair's bound proper
	if (thr_ptr) {
		thr = *thr_ptr;
		// nothing is labelled at or above twice the cap: a pair that needs such a label is not finished in this phase anyway
		// (exactness: relax_batches_bidir), and an expansion walks the few edges under 2 C - label instead of its whole list
		if (other) my_bound = min(my_bound, 2 * thr);
	}
	long long my_mu = 0x7FFFFFFFFFFFFFFFll; // (bidirectional) smallest label + other side's label this wavefront's lane has seen
	bool my_alive = false;
	u64 edges = 0;
	u32 expanded = 0;
	long long min_def = 0x7FFFFFFFFFFFFFFFll;
	int pend_q = 0, pend_t = 0; // wave-uniform
	int *const pq = s_pq[wib];
	int *const pt = s_pt[wib];

	// relax the edges [b, e) of v (label row dvb, lanes `mine`)
	auto walk = [&](int v, int64_t dvb, bool mine, int64_t b, int64_t e) {
		if (!sorted) edges += (u64)(e - b);
		bool stop = false;
		for (int64_t k0 = b; k0 < e && !stop; k0 += UNR) {
			int nn[UNR];
			T ww[UNR];
			int64_t cand[UNR], curv[UNR];
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				const bool in = k0 + u < e;
				nn[u] = in ? adj[k0 + u] : v;
				ww[u] = in ? w[k0 + u] : T(0);
				// labels are compared as bit patterns: a NaN sum with the sign bit set would look like an improvement.  A NaN
				// weight never relaxes an edge in the reference (`dist + w < dist[n]` is false): +inf behaves the same here
				if constexpr (std::is_same<T, double>::value) ww[u] = ww[u] == ww[u] ? ww[u] : __longlong_as_double(0x7FF0000000000000ll);
			}
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				if constexpr (std::is_same<T, double>::value) cand[u] = __double_as_longlong(__longlong_as_double(dvb) + ww[u]);
				else cand[u] = dvb + (int64_t)ww[u];
#if !PGQ_RELAX_NVFIRST
#if PGQ_RELAX_SEGCOND
				curv[u] = Lab<DT>::widen(dist[mine && (!sorted || cand[u] < my_bound) ? (size_t)nn[u] * LC + lane : (size_t)v * LC], Inf<T>::bits);
#else
				curv[u] = Lab<DT>::widen(dist[(size_t)nn[u] * LC + lane], Inf<T>::bits); // unconditional: all eight rows are in flight together
#endif
#endif
			}
			// how many of the eight count: up to the end of the list, the cap of the phase, or the first edge no lane's
			// candidate gets under its bound with (the list ascends by weight: no later one can either)
			int nv = 0;
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				if (stop || k0 + u >= e) break;
				if (sorted && (ww[u] > wcap || !__any(mine && cand[u] < my_bound))) {
					stop = true;
					break;
				}
				nv++;
			}
#if PGQ_RELAX_NVFIRST
			if (nv == 0) break;
#pragma unroll
			for (int u = 0; u < UNR; u++) { // past nv (and, with SEGCOND, in lanes that cannot improve): v's first word, a cached segment
#if PGQ_RELAX_SEGCOND
				const bool need = u < nv && mine && (!sorted || cand[u] < my_bound);
#else
				const bool need = u < nv;
#endif
				curv[u] = Lab<DT>::widen(dist[need ? (size_t)nn[u] * LC + lane : (size_t)v * LC], Inf<T>::bits);
			}
#endif
			if (sorted) edges += (u64)nv;
			if (other) { // an edge the cap cut (not the pair's bound): the lane's side is not exhausted, and the next cap must reach it
#pragma unroll
				for (int u = 0; u < UNR; u++) {
					if (k0 + u < e && mine && cand[u] >= my_bound && cand[u] < mu_b) {
						my_alive = true;
						min_def = min(min_def, (long long)(cand[u] >> 1));
					}
				}
			}
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				const bool go = u < nv && mine && (!sorted || cand[u] < my_bound) && cand[u] < curv[u];
				curv[u] = cand[u]; // "not improved" unless the atomic says otherwise
				if (go) curv[u] = Lab<DT>::min_into(&dist[(size_t)nn[u] * LC + lane], cand[u], Inf<T>::bits);
			}
			u64 my_im = 0;
			int my_n = 0;
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				const u64 im = __ballot(cand[u] < curv[u]);
				if (lane == u) {
					my_im = im;
					my_n = nn[u];
				}
			}
			if (!__any(my_im != 0)) continue;
			bool fresh_q = false, fresh_t = false;
			if (my_im) { // lanes 0..7, one improved neighbour each
				fresh_q = atomicOr(&dirty_nxt[my_n], my_im) == 0ull;
				fresh_t = tflag[my_n] != tepoch && atomicExch(&tflag[my_n], tepoch) != tepoch;
			}
			relax_append(pq, pend_q, fresh_q, my_n, qnxt, nq_nxt, lane);
			relax_append(pt, pend_t, fresh_t, my_n, touched, tcount, lane);
		}
	};

	if (heavy_pass) {
		for (u32 i = wave; i < nq; i += nwaves) {
			const u32 ent = hmap[i];
			const int v = hv[ent];
			const int64_t b = off[v] + (int64_t)(i - hstart[ent]) * kHeavyChunk;
			const int64_t e = min(b + kHeavyChunk, off[v + 1]);
			if (sorted && w[b] > wcap) continue; // the whole chunk lies above this phase's cap
			// the label may have improved since pass 0 queued v: any label is the length of a real path, the smaller the better
			const int64_t dvb = Lab<DT>::widen(dist[(size_t)v * LC + lane], Inf<T>::bits);
			walk(v, dvb, (hmask[ent] >> lane) & 1ull, b, e);
		}
	} else {
		// software pipeline over the wavefront's vertices: queue entry two ahead, header one ahead
		u32 i = wave, i1 = wave + nwaves, i2 = wave + 2 * nwaves;
		int v = i < nq ? qcur[i] : 0;
		int v1 = i1 < nq ? qcur[i1] : 0;
		u64 mask = 0;
		int64_t dvb = 0, b = 0, e = 0, ob = Inf<T>::bits;
		if (i < nq) {
			mask = dirty_cur[v];
			dvb = Lab<DT>::widen(dist[(size_t)v * LC + lane], Inf<T>::bits);
			if (other) ob = Lab<DT>::widen(other[(size_t)v * LC + lane], Inf<T>::bits);
			b = off[v];
			e = off[v + 1];
		}
		while (i < nq) {
			u64 mask1 = 0;
			int64_t dvb1 = 0, b1 = 0, e1 = 0, ob1 = Inf<T>::bits;
			if (i1 < nq) {
				mask1 = dirty_cur[v1];
				dvb1 = Lab<DT>::widen(dist[(size_t)v1 * LC + lane], Inf<T>::bits);
				if (other) ob1 = Lab<DT>::widen(other[(size_t)v1 * LC + lane], Inf<T>::bits);
				b1 = off[v1];
				e1 = off[v1 + 1];
			}
			const int v2 = i2 < nq ? qcur[i2] : 0;
			if (lane == 0) dirty_cur[v] = 0; // consumed; nobody else touches dirty_cur this round
			const bool live = ((mask >> lane) & 1ull) && dvb < my_bound;
			const bool mine = live && dvb < thr;
			if (other && mine && ob != Inf<T>::bits) my_mu = min(my_mu, (long long)(dvb + ob)); // (int64 weights only: exact in any order)
			const u64 deferred = __ballot(live && !mine);
			if (deferred) { // stays dirty for a later round
				my_alive |= live && !mine;
				if (live && !mine) min_def = min(min_def, (long long)dvb);
				bool fresh = false;
				if (lane == 0) fresh = atomicOr(&dirty_nxt[v], deferred) == 0ull;
				relax_append(pq, pend_q, fresh, v, qnxt, nq_nxt, lane);
			}
			const u64 mine_mask = __ballot(mine);
			if (mine_mask) {
				expanded++;
				bool is_heavy = false;
				if (heavy_ctr && e - b > kHeavyMin) is_heavy = !sorted || !(w[b + kHeavyMin] > wcap);
				if (is_heavy && other && sorted) { // no cap on the edge weight here: heavy only if the 129th edge can still get a lane under its bound
					if constexpr (!std::is_same<T, double>::value) is_heavy = __any(mine && dvb + (int64_t)w[b + kHeavyMin] < my_bound);
				}
				if (is_heavy) {
					const u32 nch = (u32)((e - b + kHeavyChunk - 1) / kHeavyChunk);
					u64 t = 0;
					if (lane == 0) t = atomicAdd(heavy_ctr, (1ull << 32) | (u64)nch);
					const u32 cs = (u32)__builtin_amdgcn_readfirstlane((int)(u32)t);
					const u32 ent = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(t >> 32));
					if (lane == 0) {
						hv[ent] = v;
						hmask[ent] = mine_mask;
						hstart[ent] = cs;
					}
					for (u32 c = lane; c < nch; c += 64) hmap[cs + c] = ent;
				} else {
					walk(v, dvb, mine, b, e);
				}
			}
			i = i1;
			i1 = i2;
			i2 += nwaves;
			v = v1;
			v1 = v2;
			mask = mask1;
			dvb = dvb1;
			ob = ob1;
			b = b1;
			e = e1;
		}
	}
	if (other) { // (wave-uniform)
		if (my_alive) alive[lane] = 1u; // plain stores of one value: no ordering needed
		if (my_mu < mu_b) atomicMin(&mu[lane], my_mu); // rare: a lane's bound improves a handful of times per search
	}
	for (int o = 32; o > 0; o >>= 1) min_def = min(min_def, (long long)__shfl_xor(min_def, o));
	if (lane == 0) {
		if (edges) atomicAdd(&s_edges, edges);
		if (expanded) atomicAdd(&s_expanded, expanded);
		if (min_def != 0x7FFFFFFFFFFFFFFFll) atomicMin(&s_min_def, min_def);
		s_pend[0][wib] = pend_q;
		s_pend[1][wib] = pend_t;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const u32 tq = (u32)(s_pend[0][0] + s_pend[0][1] + s_pend[0][2] + s_pend[0][3]);
		const u32 tt = (u32)(s_pend[1][0] + s_pend[1][1] + s_pend[1][2] + s_pend[1][3]);
		s_base[0] = tq ? atomicAdd(nq_nxt, tq) : 0u;
		s_base[1] = tt ? atomicAdd(tcount, tt) : 0u;
		if (s_edges) atomicAdd(relaxed_edges, s_edges);
		if (s_expanded) atomicAdd(relaxed_vertices, s_expanded);
		if (s_min_def != 0x7FFFFFFFFFFFFFFFll) atomicMin(min_deferred, s_min_def);
	}
	__syncthreads();
	u32 oq = s_base[0], ot = s_base[1];
	for (int k = 0; k < wib; k++) {
		oq += (u32)s_pend[0][k];
		ot += (u32)s_pend[1][k];
	}
	if (lane < pend_q) qnxt[oq + lane] = pq[lane];
	if (lane < pend_t) touched[ot + lane] = pt[lane];
}

// bound[lane] = the largest tentative label among the lane's destinations (rows [lo, hi) are sorted by lane); INF while
// one of them is unlabelled.  Workgroup-local maxima in LDS, one atomicMax per lane and workgroup.
template <typename DT>
__global__ __launch_bounds__(256) void k_lane_bounds(int64_t lo, int64_t hi, const u32 *__restrict__ skey,
                                                     const int32_t *__restrict__ sdst, u32 base_lane,
                                                     const DT *__restrict__ dist, int64_t inf_bits, long long *__restrict__ bound) {
	__shared__ long long s_b[64];
	if (threadIdx.x < 64) s_b[threadIdx.x] = 0;
	__syncthreads();
	for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
		const u32 l = skey[i] - base_lane;
		const long long d = (long long)Lab<DT>::widen(dist[(size_t)sdst[i] * LC + l], inf_bits);
		if (d > s_b[l]) atomicMax(&s_b[l], d);
	}
	__syncthreads();
	if (threadIdx.x < 64 && s_b[threadIdx.x] > 0) atomicMax(&bound[threadIdx.x], s_b[threadIdx.x]);
}

struct RelaxCounters {
	u32 nq[2];
	u32 tcount;
	u32 rounds; // rounds executed by the last k_relax_small launch
	u64 relaxed_edges;
	long long min_deferred; // smallest label (bit pattern) a round left for later (k_relax's threshold)
	u32 relaxed_vertices;   // vertices a round expanded
	u32 pad;
	u64 heavy;              // (entries << 32) | chunks of the round's heavy list (k_relax)
	long long bound[64];    // per lane: the largest tentative label among its destinations (nothing beyond it matters)
};

// the start of a round: the next queue's count, the round's statistics, the lanes' bounds
__global__ void k_round_reset(RelaxCounters *__restrict__ rc, int next_par) {
	const int t = threadIdx.x;
	if (t < 64) rc->bound[t] = 0;
	if (t == 0) {
		rc->nq[next_par] = 0;
		rc->relaxed_vertices = 0;
		rc->min_deferred = 0x7F7F7F7F7F7F7F7Fll; // > every label
		rc->heavy = 0;
	}
}

#define PGQ_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PGQ_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Small-frontier rounds without the host: ONE 1024-thread workgroup keeps relaxing round after round while the
// queue of changed vertices stays <= `limit` (a forest / long chain advances one hop per round: 18 host round
// trips per batch otherwise).  Everything another wavefront may have changed is read with agent-scope atomic
// loads (distances are updated by L2 atomics, a plain load could hit a stale L1 line); rounds are separated by
// __syncthreads().  Same relaxation and bookkeeping as k_relax.
template <typename T>
__global__ __launch_bounds__(1024) void k_relax_small(const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                      const T *__restrict__ w, int64_t *__restrict__ dist,
                                                      u64 *__restrict__ dirty0, u64 *__restrict__ dirty1,
                                                      int32_t *__restrict__ q0, int32_t *__restrict__ q1,
                                                      RelaxCounters *__restrict__ rc, int par, u32 limit,
                                                      u32 *__restrict__ qflag, u32 epoch0, u32 *__restrict__ tflag,
                                                      u32 tepoch, int32_t *__restrict__ touched, int max_rounds) {
	__shared__ u32 s_nq;
	const int lane = threadIdx.x & 63;
	const u32 wib = threadIdx.x >> 6;
	const u32 nw = blockDim.x >> 6;
	u64 edges = 0;
	int round = 0;
	for (; round < max_rounds; round++) {
		if (threadIdx.x == 0) s_nq = PGQ_LD(&rc->nq[par]);
		__syncthreads();
		const u32 nq = s_nq;
		if (nq == 0 || nq > limit) break;
		if (threadIdx.x == 0) PGQ_ST(&rc->nq[par ^ 1], 0u);
		__syncthreads();
		u64 *dirty_cur = par ? dirty1 : dirty0, *dirty_nxt = par ? dirty0 : dirty1;
		const int32_t *qcur = par ? q1 : q0;
		int32_t *qnxt = par ? q0 : q1;
		const u32 epoch = epoch0 + (u32)round;
		for (u32 i = wib; i < nq; i += nw) {
			const int v = PGQ_LD(&qcur[i]);
			u64 mask = 0;
			if (lane == 0) mask = atomicExch(&dirty_cur[v], 0ull);
			mask = __shfl(mask, 0);
			const bool mine = (mask >> lane) & 1ull;
			const int64_t dvb = PGQ_LD(&dist[(size_t)v * LC + lane]);
			const int64_t b = off[v], e = off[v + 1];
			edges += (u64)(e - b);
			for (int64_t k = b; k < e; k++) {
				const int n = adj[k];
				T wt = w[k];
				if constexpr (std::is_same<T, double>::value) wt = wt == wt ? wt : __longlong_as_double(0x7FF0000000000000ll); // NaN: like k_relax
				bool improved = false;
				if (mine) {
					int64_t cand;
					if constexpr (std::is_same<T, double>::value) cand = __double_as_longlong(__longlong_as_double(dvb) + wt);
					else cand = dvb + (int64_t)wt;
					int64_t *dp = &dist[(size_t)n * LC + lane];
					if (cand < PGQ_LD(dp)) {
						const int64_t old = atomicMin((long long *)dp, (long long)cand);
						improved = cand < old;
					}
				}
				const u64 imask = __ballot(improved);
				if (imask && lane == 0) {
					atomicOr(&dirty_nxt[n], imask);
					if (atomicExch(&qflag[n], epoch) != epoch) PGQ_ST(&qnxt[atomicAdd(&rc->nq[par ^ 1], 1u)], n);
					if (atomicExch(&tflag[n], tepoch) != tepoch) touched[atomicAdd(&rc->tcount, 1u)] = n;
				}
			}
		}
		__syncthreads();
		par ^= 1;
	}
	if (lane == 0 && edges) atomicAdd(&rc->relaxed_edges, edges);
	if (threadIdx.x == 0) rc->rounds = (u32)round;
}

// results: out[row] = dist[dst][lane]; INF -> invalid.  Also trivial rows.
template <typename DT>
__global__ void k_cheapest_results(int64_t lo, int64_t hi, const u32 *__restrict__ skey, const u32 *__restrict__ sidx,
                                   const int32_t *__restrict__ sdst, u32 base_lane, const DT *__restrict__ dist,
                                   int64_t inf_bits, int64_t *__restrict__ out, uint8_t *__restrict__ ok) {
	int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= hi) return;
	const u32 l = skey[i] - base_lane;
	const int64_t d = Lab<DT>::widen(dist[(size_t)sdst[i] * LC + l], inf_bits);
	const u32 row = sidx[i];
	if (d == inf_bits) {
		ok[row] = 0;
		out[row] = 0;
	} else {
		ok[row] = 1;
		out[row] = d;
	}
}

__global__ void k_cheapest_tails(int64_t lo_trivial, int64_t lo_null, int64_t n, const u32 *__restrict__ sidx,
                                 int64_t *__restrict__ out, uint8_t *__restrict__ ok) {
	int64_t i = lo_trivial + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u32 row = sidx[i];
	out[row] = 0; // int64 0 and +0.0 share the bit pattern
	ok[row] = i < lo_null ? 1 : 0;
}

template <typename DT>
__global__ void k_reset_touched(const int32_t *__restrict__ touched, const u32 *__restrict__ tcount,
                                DT *__restrict__ dist, DT inf_bits) {
	const u32 nt = *tcount;
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; t < (int64_t)nt * LC; t += stride) dist[(size_t)touched[t / LC] * LC + (t % LC)] = inf_bits;
}


// ---- chain pre-pass ----------------------------------------------------------------------------------------------------
// While every vertex on the way has exactly one out-edge, the set of vertices reachable from src IS that chain: if dst
// is on it the only path's cost is the answer (folded left to right like `dist[v] + w`, cheapest_path_length.cpp:29-36,
// so int64 and double results are bit-identical), if the chain ends in a vertex without out-edges first, dst is
// unreachable (NULL, :74-85).  Message-reply forests (BASELINE configs[4]) are answered entirely here; a row whose
// chain reaches a vertex with several out-edges (or runs for `cap` steps: a cycle) is left to the batched relaxation.
// One thread per row; ok = 2 marks "open".
template <typename T>
__global__ void k_chain_walk(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst, int64_t V,
                             const int64_t *__restrict__ off, const int32_t *__restrict__ adj, const T *__restrict__ w,
                             T *__restrict__ out, uint8_t *__restrict__ ok, int cap, u32 *__restrict__ counters) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const T sentinel = std::numeric_limits<T>::max() / 2;
	int step = 0;
	if (i < n) {
		const int64_t s = src[i], d = dst[i];
		T acc = (T)0;
		uint8_t state = 0; // 0 NULL, 1 found, 2 open
		if (s >= V || (s >= 0 && (d < 0 || d >= V))) {
			counters[1] = 1; // id out of range
		} else if (s >= 0) {
			int64_t x = s;
			for (;; step++) {
				if (x == d) {
					state = 1;
					break;
				}
				const int64_t b = off[x], deg = off[x + 1] - b;
				if (deg == 0) break;
				if (deg > 1 || step >= cap) {
					state = 2;
					break;
				}
				// the reference relaxes `dist[v] + w < dist[n]` against the max/2 sentinel (cheapest_path_length.cpp:15,29-36):
				// a sum that does not get under it (inf or NaN weights, sums past max/2) never labels the next vertex, and
				// nothing else leads on from a vertex with one out-edge
				const T nacc = acc + w[b];
				if (!(nacc < sentinel)) break; // state stays 0: NULL
				acc = nacc;
				x = adj[b];
			}
		}
		out[i] = state == 1 ? acc : (T)0;
		ok[i] = state;
	}
	// statistics: open rows and chain steps, one atomic each per wave
	const u64 open = __ballot(i < n && ok[i] == 2);
	int steps = step;
	for (int sh = 32; sh; sh >>= 1) steps += __shfl_xor(steps, sh);
	if ((threadIdx.x & 63) == 0) {
		if (open) atomicAdd(&counters[0], (u32)__popcll(open));
		if (steps) atomicAdd(&counters[3], (u32)steps);
	}
}
__global__ void k_collect_open_rows(int64_t n, const uint8_t *__restrict__ ok, const int64_t *__restrict__ src,
                                    const int64_t *__restrict__ dst, int64_t *__restrict__ dsrc, int64_t *__restrict__ ddst,
                                    u32 *__restrict__ didx, u32 *__restrict__ count) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool open = i < n && ok[i] == 2;
	const u64 m = __ballot(open);
	if (!m) return;
	const int lane = threadIdx.x & 63;
	u32 base = 0;
	if (lane == 0) base = atomicAdd(count, (u32)__popcll(m));
	base = __shfl(base, 0);
	if (open) {
		const u32 p = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
		dsrc[p] = src[i];
		ddst[p] = dst[i];
		didx[p] = (u32)i;
	}
}
__global__ void k_apply_open_rows(int64_t nd, const u32 *__restrict__ didx, const int64_t *__restrict__ dval,
                                  const uint8_t *__restrict__ dok, int64_t *__restrict__ out, uint8_t *__restrict__ ok) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nd) return;
	out[didx[j]] = dval[j];
	ok[didx[j]] = dok[j];
}

// ---- weighted pairs on general graphs: bidirectional band-wise label correcting per row (k_wbibfs, int64 weights) ------
// The batched relaxation below computes, per lane, the distances from the source to EVERY vertex it can reach: on the
// weighted knows graph that is the whole graph per distinct source (8.6 visits per edge, 725 pairs/s in round 1).  A
// single pair needs far less: the forward ball of src and the backward ball of dst, each of about half the distance.
// One 1024-thread workgroup per row runs delta-stepping from both ends (a row-private distance array per side in global
// memory, reset through the list of touched vertices; near / far vertex queues per side):
//     expand the side with the smaller radius by one band [r, r + delta): relax the near queue to its fixpoint (a vertex
//     improved to below r + delta goes back into the near queue, others into the far queue); every improvement of a vertex
//     the other side has labelled offers  best = min(best, new label + other side's label)
//     the band is complete: r += delta; stop when r_fwd + r_bwd >= best; refill the near queue from the far queue
//     (skipping bands without vertices); an empty far queue means that side's closure is exhausted
// Exactness (int64 sums, any order): when r_fwd + r_bwd >= best > D were to hold, take the last vertex x of an optimal path
// with d_fwd(x) < r_fwd and its successor y: d_bwd(y) = D - d_fwd(y) <= D - r_fwd < r_bwd, so both are final, and the later
// of "x relaxed forward" / "y relaxed backward" has offered d_fwd(x) + w + d_bwd(y) = D.  Checked against Dijkstra in a
// Python model of exactly this schedule (zero weights, band skipping, duplicates in the queues) before it was written here.
// A row whose queues or work exceed their caps stays open (ok = 2) and goes to the batched relaxation.
constexpr long long kWbInf = 1LL << 62;
struct WbCounters {
	u32 open;
	u32 pad;
	unsigned long long relaxed;
};

__global__ void k_iota32(int64_t n, u32 *__restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = (u32)i;
}
__global__ void k_gather_weights(int64_t n, const u32 *__restrict__ slot, const int64_t *__restrict__ w, int64_t *__restrict__ rw) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) rw[i] = w[slot[i]];
}

// adjacency walk of a vertex list with each vertex's 64-bit payload (its own label) and the slot position of a lane's four
// entries (for the weights): g(v, valid, payload, t)
template <typename G>
__device__ __forceinline__ unsigned long long wb_walk(const u32 *__restrict__ list, int list_n, int w, int stride,
                                                      const int64_t *__restrict__ xoff, const int32_t *__restrict__ xadj,
                                                      const long long *__restrict__ pay, G g) {
	const int lane = threadIdx.x & 63;
	unsigned long long entries = 0;
	for (int pb = w; pb < list_n; pb += 64 * stride) {
		int vb = 0, ve = 0;
		long long pv = 0;
		const int p = pb + lane * stride;
		if (p < list_n) {
			const u32 vid = list[p];
			vb = (int)xoff[vid];
			ve = (int)xoff[vid + 1];
			pv = __hip_atomic_load(&pay[vid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		const int cnt = min(64, (list_n - pb + stride - 1) / stride);
		int j = -1, q = 0, e = 0, b = 0;
		long long cp = 0;
		auto seek = [&]() {
			for (j++; j < cnt; j++) {
				b = __builtin_amdgcn_readlane(vb, j);
				e = __builtin_amdgcn_readlane(ve, j);
				if (e > b) {
					q = b & ~3;
					const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)pv, j);
					const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)((unsigned long long)pv >> 32), j);
					cp = (long long)(((unsigned long long)hi << 32) | lo);
					return;
				}
			}
		};
		seek();
		while (j < cnt) { // one chunk at a time: the callback's own memory operations are the long part
			const int t = q + 4 * lane;
			typedef int v4i __attribute__((ext_vector_type(4)));
			const v4i r = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(xadj + (t < e ? t : q)));
			const u32 valid = (u32)(t >= b && t < e) | ((u32)(t + 1 >= b && t + 1 < e) << 1) |
			                  ((u32)(t + 2 >= b && t + 2 < e) << 2) | ((u32)(t + 3 >= b && t + 3 < e) << 3);
			entries += (unsigned long long)(min(e, q + 256) - max(b, q));
			const long long cur = cp;
			q += 256;
			if (q >= e) seek();
			g(make_int4(r.x, r.y, r.z, r.w), valid, cur, t);
		}
	}
	return entries;
}

__global__ __launch_bounds__(1024) void k_wbibfs(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                 const u32 *__restrict__ didx, const int64_t *__restrict__ off,
                                                 const int32_t *__restrict__ adj, const int64_t *__restrict__ w,
                                                 const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                                 const int64_t *__restrict__ rw, int64_t V, long long delta,
                                                 long long work_cap, int qcap, int fcap, int prune, long long *__restrict__ dist_all,
                                                 u32 *__restrict__ queues_all, int64_t *__restrict__ out,
                                                 uint8_t *__restrict__ ok, WbCounters *__restrict__ wc) {
	// per workgroup: dist[side][V + 1] (entry V takes the masked lanes), near[side][parity][qcap] (vertices inside the
	// current band, duplicates possible), far[side][parity][fcap] and touched[side][fcap] (every labelled vertex once)
	long long *const dist0 = dist_all + (size_t)blockIdx.x * 2 * (size_t)(V + 1);
	u32 *const qb = queues_all + (size_t)blockIdx.x * (4 * (size_t)qcap + 6 * (size_t)fcap);
	u32 *const fb = qb + 4 * (size_t)qcap;  // far[side][parity]
	u32 *const tb = fb + 4 * (size_t)fcap;  // touched[side]
	__shared__ u32 s_nn, s_nfar[2], s_nt[2], s_cnt2[2];
	__shared__ unsigned long long s_best, s_min, s_work;
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	unsigned long long relaxed = 0;
	u32 open_rows = 0;
	for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
		__syncthreads();
		const int64_t s = src[i], d = dst[i]; // open rows of the chain pre-pass: ids in range, src != dst
		const u32 row = didx[i];
		if (tid == 0) {
			s_nfar[0] = s_nfar[1] = 0;
			s_nt[0] = s_nt[1] = 1;
			s_best = (unsigned long long)kWbInf;
			s_work = 0;
			__hip_atomic_store(&dist0[s], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&dist0[(V + 1) + d], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			tb[0] = (u32)s;
			tb[(size_t)fcap] = (u32)d;
			qb[0] = (u32)s;
			qb[2 * (size_t)qcap] = (u32)d;
		}
		u32 nn0 = 1, nn1 = 1;      // near-queue fill per side
		int par0 = 0, par1 = 0;    // which of a side's two near buffers is current
		int fpar0 = 0, fpar1 = 0;  // ... and of its two far buffers
		long long r0 = 0, r1 = 0;  // completed radius per side
		int state = 2;             // 0 NULL, 1 found, 2 open
		__threadfence_block();
		__syncthreads();
		if (s == d) state = 1, s_best = 0; // not expected here; kept for safety (every thread writes the same value)
		while (state == 2 && s != d) {
			const int side = r0 <= r1 ? 0 : 1;
			const int64_t *xoff = side ? roff : off;
			const int32_t *xadj = side ? radj : adj;
			const int64_t *xw = side ? rw : w;
			long long *mine = dist0 + (size_t)side * (size_t)(V + 1);
			const long long *theirs = dist0 + (size_t)(side ^ 1) * (size_t)(V + 1);
			const int fpar = side ? fpar1 : fpar0;
			u32 *far = fb + (size_t)(side * 2 + fpar) * fcap;
			u32 *touched = tb + (size_t)side * fcap;
			const long long rn = (side ? r1 : r0) + delta;
			u32 nn = side ? nn1 : nn0;
			int par = side ? par1 : par0;
			bool over = false;
			// -- the band [r, rn): relax the near queue to its fixpoint
			while (nn > 0) {
				const u32 *cur = qb + (size_t)(side * 2 + par) * qcap;
				u32 *nxt = qb + (size_t)(side * 2 + (par ^ 1)) * qcap;
				if (tid == 0) s_nn = 0;
				__syncthreads();
				unsigned long long lbest = (unsigned long long)kWbInf;
				const long long r_other = side ? r0 : r1;
				const unsigned long long e2 = wb_walk(cur, (int)nn, wib, 16, xoff, xadj, mine, [&](const int4 &v, u32 valid, long long dv, int t) {
					// optional pruning (wbibfs_prune): a vertex whose label plus the other side's completed radius already
					// reaches `best` cannot start a better path — an out-neighbour u the other side has settled with
					// w + d_other(u) < r_other makes this vertex settled there too, and its own labelling offered that sum
					if (prune && (unsigned long long)(dv + r_other) >= *(volatile unsigned long long *)&s_best) valid = 0;
					const u32 xs[4] = { (u32)v.x, (u32)v.y, (u32)v.z, (u32)v.w };
					u32 x[4];
					long long nd[4], old[4], oth[4];
					// no per-lane branches around memory operations: masked lanes use the spare entry V with an infinite label
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const bool in = (valid >> k) & 1u;
						x[k] = in ? xs[k] : (u32)V;
						const long long wt = xw[in ? (size_t)(t + k) : (size_t)0]; // masked lanes read slot 0
						nd[k] = in ? dv + wt : kWbInf;
					}
#pragma unroll
					for (int k = 0; k < 4; k++) old[k] = atomicMin(&mine[x[k]], nd[k]);
#pragma unroll
					for (int k = 0; k < 4; k++) oth[k] = __hip_atomic_load(&theirs[x[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const bool in = (valid >> k) & 1u;
						const bool better = in && nd[k] < old[k];
						const bool first = better && old[k] >= kWbInf;
						if (better && oth[k] < kWbInf) lbest = min(lbest, (unsigned long long)(nd[k] + oth[k]));
						// a vertex enters the far queue once, when it is first labelled beyond the band: a later improvement either
						// stays beyond the band (its entry is still there) or moves it into the near queue
						const bool to_near = better && nd[k] < rn, to_far = first && !(nd[k] < rn);
						const u64 mn = __ballot(to_near), mf = __ballot(to_far), mt = __ballot(first);
						if (mn) {
							u32 base = 0;
							if (lane == 0) base = atomicAdd(&s_nn, (u32)__popcll(mn));
							base = (u32)__shfl((int)base, 0);
							const u32 slot = base + __builtin_amdgcn_mbcnt_hi((u32)(mn >> 32), __builtin_amdgcn_mbcnt_lo((u32)mn, 0u));
							if (to_near && slot < (u32)qcap) nxt[slot] = x[k];
						}
						if (mf) {
							u32 base = 0;
							if (lane == 0) base = atomicAdd(&s_nfar[side], (u32)__popcll(mf));
							base = (u32)__shfl((int)base, 0);
							const u32 slot = base + __builtin_amdgcn_mbcnt_hi((u32)(mf >> 32), __builtin_amdgcn_mbcnt_lo((u32)mf, 0u));
							if (to_far && slot < (u32)fcap) far[slot] = x[k];
						}
						if (mt) {
							u32 base = 0;
							if (lane == 0) base = atomicAdd(&s_nt[side], (u32)__popcll(mt));
							base = (u32)__shfl((int)base, 0);
							const u32 slot = base + __builtin_amdgcn_mbcnt_hi((u32)(mt >> 32), __builtin_amdgcn_mbcnt_lo((u32)mt, 0u));
							if (first && slot < (u32)fcap) touched[slot] = x[k];
						}
					}
				});
				for (int o = 32; o > 0; o >>= 1) {
					const unsigned long long y = __shfl_xor(lbest, o);
					lbest = y < lbest ? y : lbest;
				}
				if (lane == 0) {
					if (lbest < (unsigned long long)kWbInf) atomicMin(&s_best, lbest);
					if (e2) atomicAdd(&s_work, e2);
				}
				__syncthreads();
				nn = s_nn;
				par ^= 1;
				if (nn > (u32)qcap || s_nfar[side] > (u32)fcap || s_nt[side] > (u32)fcap || (long long)s_work > work_cap) {
					over = true;
					break;
				}
				__syncthreads(); // s_nn is reset by the next round
			}
			if (over) break; // state stays 2: the row is left to the batched relaxation
			if (side) r1 = rn, nn1 = 0, par1 = par;
			else r0 = rn, nn0 = 0, par0 = par;
			if ((unsigned long long)(r0 + r1) >= s_best) {
				state = 1;
				break;
			}
			// -- refill the near queue from the far queue; bands without vertices are skipped
			const u32 nf = s_nfar[side];
			const long long rr = side ? r1 : r0;
			if (tid == 0) s_min = (unsigned long long)kWbInf;
			__syncthreads();
			unsigned long long lm = (unsigned long long)kWbInf;
			for (u32 p = tid; p < nf; p += 1024) {
				const long long dv = __hip_atomic_load(&mine[far[p]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (dv >= rr) lm = min(lm, (unsigned long long)dv); // below rr: settled in an earlier band (a stale copy)
			}
			for (int o = 32; o > 0; o >>= 1) {
				const unsigned long long y = __shfl_xor(lm, o);
				lm = y < lm ? y : lm;
			}
			if (lane == 0 && lm < (unsigned long long)kWbInf) atomicMin(&s_min, lm);
			__syncthreads();
			const unsigned long long m = s_min;
			if (m >= (unsigned long long)kWbInf) { // nothing left on this side: its closure is complete
				state = s_best < (unsigned long long)kWbInf ? 1 : 0;
				break;
			}
			long long base_r = rr;
			if ((long long)m >= rr + delta) base_r = ((long long)m / delta) * delta;
			if (side) r1 = base_r;
			else r0 = base_r;
			if ((unsigned long long)(r0 + r1) >= s_best) {
				state = 1;
				break;
			}
			// the side's near buffers are free now: [par] becomes the new near queue; what stays far goes to the other far buffer
			u32 *newnear = qb + (size_t)(side * 2 + par) * qcap;
			u32 *keep = fb + (size_t)(side * 2 + (fpar ^ 1)) * fcap;
			if (tid == 0) s_cnt2[0] = s_cnt2[1] = 0;
			__syncthreads();
			for (u32 p0 = 0; p0 < nf; p0 += 1024) {
				const u32 p = p0 + tid;
				u32 u = 0;
				long long dv = -1;
				if (p < nf) {
					u = far[p];
					dv = __hip_atomic_load(&mine[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				const bool live = p < nf && dv >= base_r;
				const bool tn = live && dv < base_r + delta, tk = live && !tn;
				const u64 mn = __ballot(tn), mk = __ballot(tk);
				if (mn) {
					u32 b2 = 0;
					if (lane == 0) b2 = atomicAdd(&s_cnt2[0], (u32)__popcll(mn));
					b2 = (u32)__shfl((int)b2, 0);
					const u32 slot = b2 + __builtin_amdgcn_mbcnt_hi((u32)(mn >> 32), __builtin_amdgcn_mbcnt_lo((u32)mn, 0u));
					if (tn && slot < (u32)qcap) newnear[slot] = u;
				}
				if (mk) {
					u32 b2 = 0;
					if (lane == 0) b2 = atomicAdd(&s_cnt2[1], (u32)__popcll(mk));
					b2 = (u32)__shfl((int)b2, 0);
					if (tk) keep[b2 + __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0u))] = u;
				}
			}
			__syncthreads();
			const u32 nnew = s_cnt2[0], nkeep = s_cnt2[1];
			if (nnew > (u32)qcap) break; // more vertices in one band than the near queue holds: the row stays open
			if (tid == 0) s_nfar[side] = nkeep;
			if (side) nn1 = nnew, fpar1 = fpar ^ 1;
			else nn0 = nnew, fpar0 = fpar ^ 1;
			__threadfence_block();
			__syncthreads();
		}
		__syncthreads();
		if (tid == 0) {
			if (state == 2) open_rows++;
			else {
				out[row] = state == 1 ? (int64_t)s_best : 0;
				ok[row] = (uint8_t)state;
			}
		}
		// the labels go back to "infinite" through the touched lists (a list that overflowed: everything)
		for (int sd = 0; sd < 2; sd++) {
			long long *dd = dist0 + (size_t)sd * (size_t)(V + 1);
			const u32 nt = s_nt[sd];
			if (nt <= (u32)fcap) {
				const u32 *tl = tb + (size_t)sd * fcap;
				for (u32 p = tid; p < nt; p += 1024) dd[tl[p]] = kWbInf;
			} else {
				for (int64_t v = tid; v < V; v += 1024) dd[v] = kWbInf;
			}
			if (tid == 0) dd[V] = kWbInf;
		}
		relaxed += tid == 0 ? s_work : 0ull;
	}
	if (tid == 0) {
		if (open_rows) atomicAdd(&wc->open, open_rows);
		if (relaxed) atomicAdd(&wc->relaxed, relaxed);
	}
}


static std::mutex g_rw_lock;
// in-edge weights in reverse-CSR order + the mean weight, built on first use (a stable sort of the forward slots by
// destination: the permutation the upload used for radj)
static int ensure_reverse_weights(pgq_csr *c, Workspace *ws) {
	std::lock_guard<std::mutex> g(g_rw_lock);
	if (c->rw || c->E == 0) return PGQ_OK;
	hipStream_t st = ws->stream;
	const int64_t E = c->E;
	DevBuf iota, sslot, skey, tmp, sum;
	auto body = [&]() -> int {
		for (DevBuf *b : { &iota, &sslot, &skey }) PGQ_TRY(b->reserve((size_t)E * 4));
		PGQ_TRY(sum.reserve(64));
		hipLaunchKernelGGL(k_iota32, dim3(blocks_for(E)), dim3(256), 0, st, E, iota.as<u32>());
		int end_bit = 1;
		while ((1LL << end_bit) < c->V) end_bit++;
		size_t sb = 0, rb = 0;
		const u32 *keys = reinterpret_cast<const u32 *>(c->adj);
		PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, keys, skey.as<u32>(), iota.as<u32>(), sslot.as<u32>(), (int)E, 0, end_bit, st));
		PGQ_HIP_TRY(hipcub::DeviceReduce::Sum(nullptr, rb, (const int64_t *)c->w, sum.as<int64_t>(), (int)E, st));
		PGQ_TRY(tmp.reserve(std::max(sb, rb) + 16));
		PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, sb, keys, skey.as<u32>(), iota.as<u32>(), sslot.as<u32>(), (int)E, 0, end_bit, st));
		void *rw = nullptr;
		PGQ_TRY(dev_alloc(&rw, (size_t)E * 8));
		hipLaunchKernelGGL(k_gather_weights, dim3(blocks_for(E)), dim3(256), 0, st, E, sslot.as<u32>(), (const int64_t *)c->w, (int64_t *)rw);
		PGQ_HIP_TRY(hipcub::DeviceReduce::Sum(tmp.p, rb, (const int64_t *)c->w, sum.as<int64_t>(), (int)E, st));
		int64_t total = 0;
		hipError_t e1 = hipMemcpyAsync(&total, sum.p, 8, hipMemcpyDeviceToHost, st);
		hipError_t e2 = hipStreamSynchronize(st);
		if (e1 != hipSuccess || e2 != hipSuccess) {
			dev_free(rw);
			return fail(PGQ_ERR_HIP, "building the reverse weights failed");
		}
		c->w_mean = (double)total / (double)E;
		c->rw = rw;
		return PGQ_OK;
	};
	const int rc = body();
	for (DevBuf *b : { &iota, &sslot, &skey, &tmp, &sum }) b->release();
	return rc;
}

// a new phase (higher cap on the edge weight): every labelled vertex is expanded again over the longer prefix of its list
__global__ void k_redirty(const int32_t *__restrict__ touched, const u32 *__restrict__ tcount, u64 *__restrict__ dirty,
                          int32_t *__restrict__ q, u32 *__restrict__ nq) {
	const u32 n = *tcount;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int v = touched[i];
		dirty[v] = ~0ull; // lanes without a label are skipped by their bound test
		q[i] = v;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) *nq = n;
}

// the forward adjacency with every vertex's list sorted by weight (non-negative weights: their bit patterns sort like the
// values, int64 and double alike) — segmented radix sort, once per CSR — and the largest weight
// sort keys of double weights: the bit pattern orders non-negative doubles like their values, once -0.0 (accepted: it
// is not < 0, and adds like +0.0) has lost its sign and a NaN (never relaxes an edge: `dist + w < dist[n]` is false,
// here as in the reference) counts as +inf
__global__ void k_weight_keys(const unsigned long long *__restrict__ w, int64_t E, unsigned long long *__restrict__ key) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= E) return;
	key[i] = min(w[i] & 0x7FFFFFFFFFFFFFFFull, 0x7FF0000000000000ull);
}

static int ensure_weight_sorted(pgq_csr *c, Workspace *ws) {
	std::lock_guard<std::mutex> g(g_rw_lock);
	if (c->wadj || c->E == 0 || !c->w) return PGQ_OK;
	hipStream_t st = ws->stream;
	const int64_t E = c->E;
	DevBuf tmp, mx, canon;
	int32_t *wadj = nullptr;
	void *wsorted = nullptr;
	auto body = [&]() -> int {
		PGQ_TRY(dev_alloc_as(&wadj, (size_t)E + 4));
		PGQ_TRY(dev_alloc(&wsorted, (size_t)E * 8));
		PGQ_TRY(mx.reserve(64));
		size_t sb = 0, rb = 0;
		const unsigned long long *keys = (const unsigned long long *)c->w;
		if (c->w_type == PGQ_W_DOUBLE) {
			PGQ_TRY(canon.reserve((size_t)E * 8));
			hipLaunchKernelGGL(k_weight_keys, dim3(blocks_for(E)), dim3(256), 0, st, keys, E, canon.as<unsigned long long>());
			keys = canon.as<unsigned long long>();
		}
		unsigned long long *keys_out = (unsigned long long *)wsorted;
		PGQ_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, sb, keys, keys_out, c->adj, wadj, (int)E, (int)c->V, c->off,
		                                                        c->off + 1, 0, 64, st));
		PGQ_HIP_TRY(hipcub::DeviceReduce::Max(nullptr, rb, keys, mx.as<unsigned long long>(), (int)E, st));
		PGQ_TRY(tmp.reserve(std::max(sb, rb) + 16));
		PGQ_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp.p, sb, keys, keys_out, c->adj, wadj, (int)E, (int)c->V, c->off,
		                                                        c->off + 1, 0, 64, st));
		PGQ_HIP_TRY(hipcub::DeviceReduce::Max(tmp.p, rb, keys, mx.as<unsigned long long>(), (int)E, st));
		unsigned long long m = 0;
		PGQ_HIP_TRY(hipMemcpyAsync(&m, mx.p, 8, hipMemcpyDeviceToHost, st));
		PGQ_HIP_TRY(hipStreamSynchronize(st));
		c->w_max_bits = m; // the largest weight, as its bit pattern (int64: the value)
		return PGQ_OK;
	};
	const int rc = body();
	(void)hipStreamSynchronize(st);
	tmp.release();
	mx.release();
	canon.release();
	if (rc != PGQ_OK) {
		dev_free(wadj);
		dev_free(wsorted);
		return rc;
	}
	c->wsorted = wsorted;
	c->wadj = wadj;
	return PGQ_OK;
}

// mean edge weight (int64 or double), computed once per CSR: the band width of the ordered relaxation rounds
static int ensure_weight_mean(pgq_csr *c, Workspace *ws) {
	std::lock_guard<std::mutex> g(g_rw_lock);
	if (c->w_mean > 0 || c->E == 0 || !c->w) return PGQ_OK;
	hipStream_t st = ws->stream;
	DevBuf tmp, sum;
	auto body = [&]() -> int {
		PGQ_TRY(sum.reserve(64));
		size_t rb = 0;
		double total = 0;
		if (c->w_type == PGQ_W_DOUBLE) {
			PGQ_HIP_TRY(hipcub::DeviceReduce::Sum(nullptr, rb, (const double *)c->w, sum.as<double>(), (int)c->E, st));
			PGQ_TRY(tmp.reserve(rb + 16));
			PGQ_HIP_TRY(hipcub::DeviceReduce::Sum(tmp.p, rb, (const double *)c->w, sum.as<double>(), (int)c->E, st));
			PGQ_HIP_TRY(hipMemcpyAsync(&total, sum.p, 8, hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
		} else {
			int64_t t = 0;
			PGQ_HIP_TRY(hipcub::DeviceReduce::Sum(nullptr, rb, (const int64_t *)c->w, sum.as<int64_t>(), (int)c->E, st));
			PGQ_TRY(tmp.reserve(rb + 16));
			PGQ_HIP_TRY(hipcub::DeviceReduce::Sum(tmp.p, rb, (const int64_t *)c->w, sum.as<int64_t>(), (int)c->E, st));
			PGQ_HIP_TRY(hipMemcpyAsync(&t, sum.p, 8, hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
			total = (double)t;
		}
		c->w_mean = std::max(total / (double)c->E, 1e-300);
		return PGQ_OK;
	};
	const int rc = body();
	tmp.release();
	sum.release();
	return rc;
}

// rows [0, nd) in ws->def_src / def_dst / def_idx (the chain pre-pass's open rows): answered rows get their value and
// flag written into d_out / d_ok, the others keep ok = 2.  *left = rows still open.
static int weighted_pairs_prepass(pgq_csr *c, Workspace *ws, u32 nd, int64_t *d_out, uint8_t *d_ok, u32 *left) {
	*left = nd;
	const Options &opt = options();
	if (!opt.wbibfs || c->w_type != PGQ_W_INT64 || (int64_t)nd > (int64_t)opt.wbibfs_rows) return PGQ_OK;
	const int qcap = std::max(1024, opt.wbibfs_queue);
	// far / touched hold every labelled vertex once: the boundary of a ball in a graph of degree ~90 is ~90 x the ball
	const int fcap = (int)std::min<int64_t>(std::max<int64_t>(c->V, 1024), (int64_t)std::max(1024, opt.wbibfs_far));
	const size_t q_words = (size_t)4 * qcap + (size_t)6 * fcap;
	const size_t per_wg = (size_t)2 * (size_t)(c->V + 1) * 8 + q_words * 4;
	const size_t budget = (size_t)std::max(0, opt.wbibfs_mem_mb) << 20;
	const u32 grid = (u32)std::min<size_t>(std::min<u32>(nd, 256), budget / per_wg);
	if (grid == 0) return PGQ_OK;
	PGQ_TRY(ensure_reverse_weights(c, ws));
	hipStream_t st = ws->stream;
	const size_t dist_words = (size_t)grid * 2 * (size_t)(c->V + 1);
	const bool fresh = ws->wb_scratch.cap < dist_words * 8 + (size_t)grid * q_words * 4 + 64 || ws->wb_V != c->V || ws->wb_grid != (int)grid;
	PGQ_TRY(ws->wb_scratch.reserve(dist_words * 8 + (size_t)grid * q_words * 4 + 64));
	long long *dist = ws->wb_scratch.as<long long>();
	u32 *queues = reinterpret_cast<u32 *>(dist + dist_words);
	if (fresh) { // every label infinite; the kernel restores what it touched
		hipLaunchKernelGGL(k_fill64, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, (int64_t *)dist, (int64_t)dist_words, (int64_t)kWbInf);
		ws->wb_V = c->V;
		ws->wb_grid = (int)grid;
	}
	PGQ_TRY(ws->counters.reserve(sizeof(Counters)));
	WbCounters *wc = reinterpret_cast<WbCounters *>(reinterpret_cast<char *>(ws->counters.p) + 32);
	PGQ_HIP_TRY(hipMemsetAsync(wc, 0, sizeof(WbCounters), st));
	const long long delta = std::max<long long>(1, (long long)(c->w_mean / std::max(1, opt.wbibfs_delta_div)));
	{
		KernelTimer kt(st, K_RELAX);
		hipLaunchKernelGGL(k_wbibfs, dim3(grid), dim3(1024), 0, st, (int64_t)nd, ws->def_src.as<int64_t>(), ws->def_dst.as<int64_t>(),
		                   ws->def_idx.as<u32>(), c->off, c->adj, (const int64_t *)c->w, c->roff, c->radj, (const int64_t *)c->rw,
		                   c->V, delta, (long long)std::max(1, opt.wbibfs_cap), qcap, fcap, opt.wbibfs_prune, dist, queues, d_out, d_ok, wc);
		kt.stop();
	}
	WbCounters h;
	PGQ_HIP_TRY(hipMemcpyAsync(&h, wc, sizeof(h), hipMemcpyDeviceToHost, st));
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	KernelTimer::flush();
	pgq_stats_t &S = tstats().s;
	S.edges_scanned += (int64_t)h.relaxed;
	S.algo_bytes[K_RELAX] += 28.0 * (double)h.relaxed; // adjacency entry, weight, label RMW, other side's label
	S.meet_pairs += (int64_t)nd - (int64_t)h.open;
	*left = h.open;
	return PGQ_OK;
}

template <typename T>
static int cheapest_device(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                           int64_t *d_out, uint8_t *d_ok, bool chain);

template <typename T>
static int cheapest_with_chains(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                int64_t *d_out, uint8_t *d_ok) {
	hipStream_t st = ws->stream;
	PGQ_TRY(ws->counters.reserve(sizeof(Counters)));
	u32 *d_cnt = reinterpret_cast<u32 *>(ws->counters.p); // [0] open rows, [1] bad id, [2] compaction cursor
	PGQ_HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, st));
	{
		KernelTimer kt(st, K_RELAX);
		hipLaunchKernelGGL(k_chain_walk<T>, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, d_dst, c->V, c->off, c->adj,
		                   (const T *)c->w, (T *)d_out, d_ok, std::max(1, options().chain_cap), d_cnt);
		kt.stop();
	}
	u32 h[4] = { 0, 0, 0, 0 };
	PGQ_HIP_TRY(hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	KernelTimer::flush();
	{ // a chain step reads off[x], off[x+1], w[b], adj[b]; a row reads its pair and writes value + flag
		pgq_stats_t &S = tstats().s;
		S.edges_scanned += h[3];
		S.algo_bytes[K_RELAX] += 28.0 * h[3] + 25.0 * (double)n;
	}
	if (h[1]) return fail(PGQ_ERR_INVALID_ARG, "src/dst rowid out of range [0,V)");
	const u32 nd_chain = h[0];
	tstats().s.meet_pairs += n - (int64_t)nd_chain;
	if (nd_chain == 0) {
		tstats().s.pairs += n;
		return PGQ_OK;
	}
	PGQ_TRY(ws->def_src.reserve((size_t)nd_chain * 8));
	PGQ_TRY(ws->def_dst.reserve((size_t)nd_chain * 8));
	PGQ_TRY(ws->def_idx.reserve((size_t)nd_chain * 4));
	PGQ_TRY(ws->def_len.reserve((size_t)nd_chain * 8));
	PGQ_TRY(ws->def_off.reserve((size_t)nd_chain));
	hipLaunchKernelGGL(k_collect_open_rows, dim3(blocks_for(n)), dim3(256), 0, st, n, d_ok, d_src, d_dst,
	                   ws->def_src.as<int64_t>(), ws->def_dst.as<int64_t>(), ws->def_idx.as<u32>(), d_cnt + 2);
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	u32 nd = nd_chain;
	{ // general graphs: a bidirectional search per row; what it leaves open (caps) is collected again
		u32 left = nd;
		PGQ_TRY(weighted_pairs_prepass(c, ws, nd, d_out, d_ok, &left));
		if (left != nd) {
			nd = left;
			if (nd == 0) {
				tstats().s.pairs += n;
				return PGQ_OK;
			}
			PGQ_HIP_TRY(hipMemsetAsync(d_cnt + 2, 0, 4, st));
			hipLaunchKernelGGL(k_collect_open_rows, dim3(blocks_for(n)), dim3(256), 0, st, n, d_ok, d_src, d_dst,
			                   ws->def_src.as<int64_t>(), ws->def_dst.as<int64_t>(), ws->def_idx.as<u32>(), d_cnt + 2);
			PGQ_HIP_TRY(hipStreamSynchronize(st));
		}
	}
	{
		WorkspaceLease inner;
		PGQ_TRY(inner.acquire());
		PGQ_TRY(cheapest_device<T>(c, inner.ws, nd, ws->def_src.as<int64_t>(), ws->def_dst.as<int64_t>(),
		                           ws->def_len.as<int64_t>(), ws->def_off.as<uint8_t>(), false));
		hipLaunchKernelGGL(k_apply_open_rows, dim3(blocks_for(nd)), dim3(256), 0, st, (int64_t)nd, ws->def_idx.as<u32>(),
		                   ws->def_len.as<int64_t>(), ws->def_off.as<uint8_t>(), d_out, d_ok);
		PGQ_HIP_TRY(hipStreamSynchronize(st));
	}
	tstats().s.pairs += n - (int64_t)nd; // the rows of the inner call were counted there
	return PGQ_OK;
}

// Whether the relaxation of this CSR runs "light edges first" (weight-sorted lists under a doubling cap, one host round
// trip per round) or plain rounds (every edge of a changed vertex; few changed vertices loop on the device, k_relax_small).
// The cap pays by the edges it does not scan: on the weighted knows graph (89 edges per vertex, cheapest paths use the
// lightest few per cent) it took 9.3 s to 0.6 s per 4096 pairs.  A vertex with three or four edges has nothing to skip:
// a weighted ring with chords ran 1830 rounds (0.28 s) under the cap against 365 rounds (0.05 s) plain — round 4 shipped
// that regression.  relax_light = 1 (shipped): by the mean out-degree; 2: always (tests); 0: never.
static bool light_edges_first(const pgq_csr *c) {
	const Options &o = options();
	if (o.relax_light == 0 || c->E <= 0) return false;
	if (o.relax_light >= 2) return true;
	return (double)c->E >= (double)std::max(1, o.relax_light_min_degree) * (double)std::max<int64_t>(c->V, 1);
}

// The batches b0, b0 + bstride, ... of a call: `ws` holds the sorted rows and the distinct sources (read-only here),
// `priv` everything a batch writes (labels, dirty words, queues) and the stream.  Batches are independent, so several
// host threads run this side by side on their own workspaces (cheapest_device).
// Whether this CSR's labels are stored as int32 (Lab<int32_t> above): int64 weights, the light-edges-first path (the
// device-side small rounds of the plain path keep 8-byte labels), and no path sum that could reach 2^31 - 1.
template <typename T> static bool labels_fit_32(const pgq_csr *c) {
	if (!std::is_same<T, int64_t>::value || !options().relax_labels32 || !light_edges_first(c) || !c->wadj) return false;
	if (!(c->w_mean > 0 && c->w_mean < 1e300)) return false; // (plain rounds then: relax_batches' own test)
	const unsigned long long w_max = c->w_max_bits; // int64 weights are non-negative here: the bit pattern is the value
	return w_max < (1ull << 31) && (double)w_max * (double)std::max<int64_t>(c->V, 1) < 2147483000.0;
}

template <typename T, typename DT>
static int relax_batches(pgq_csr *c, Workspace *ws, Workspace *priv, int b0, int bstride, int nb, u32 U, int64_t *d_out,
                         uint8_t *d_ok, const bool light) {
	// `light` (and with it the label width DT) was decided ONCE by cheapest_device: a concurrent pgq_set_option between two
	// evaluations used to leave a worker with 4-byte labels outside the light-edges-first path (an "internal error" return)
	hipStream_t st = priv->stream;
	const int64_t V = c->V;
	const int64_t inf_bits = Inf<T>::bits;
	const DT inf_label = (DT)Lab<DT>::unlabelled(inf_bits);
	pgq_stats_t &S = tstats().s;
	const int64_t *bs = ws->h_bstart;
	const size_t cells = (size_t)std::max<int64_t>(V, 1) * LC;
	// distances start at INF; a full fill only when the array is new, afterwards only touched rows are reset
	// (the INF pattern differs between int64 and double, the cell size between 8- and 4-byte labels)
	const int type_tag = sizeof(DT) == 4 ? 3 : (std::is_same<T, double>::value ? 2 : 1);
	const bool fresh = priv->dist.cap < cells * sizeof(DT) || priv->dist_V != V || priv->dist_lanes != type_tag;
	priv->dist_V = -1; // stays invalid if we bail out half-way; restored at the end
	PGQ_TRY(priv->dist.reserve(cells * sizeof(DT)));
	for (int k = 0; k < 2; k++) PGQ_TRY(priv->dirty[k].reserve((size_t)std::max<int64_t>(V, 1) * 8));
	PGQ_TRY(priv->qbuf[0].reserve((size_t)std::max<int64_t>(V, 1) * 4));
	PGQ_TRY(priv->qbuf[1].reserve((size_t)std::max<int64_t>(V, 1) * 4));
	PGQ_TRY(priv->touched.reserve((size_t)std::max<int64_t>(V, 1) * 4));
	PGQ_TRY(priv->qflag.reserve((size_t)std::max<int64_t>(V, 1) * 4));
	PGQ_TRY(priv->tflag.reserve((size_t)std::max<int64_t>(V, 1) * 4));
	// heavy list of a round: entries (vertex, lanes, first chunk) and the chunk -> entry map (<= E/64 + V chunks)
	PGQ_TRY(priv->hv.reserve((size_t)std::max<int64_t>(V, 1) * 4));
	PGQ_TRY(priv->hmask.reserve((size_t)std::max<int64_t>(V, 1) * 8));
	PGQ_TRY(priv->hstart.reserve((size_t)std::max<int64_t>(V, 1) * 4));
	PGQ_TRY(priv->hmap.reserve((size_t)(c->E / 64 + std::max<int64_t>(V, 1)) * 4));
	PGQ_TRY(priv->counters.reserve(sizeof(Counters)));
	static_assert(sizeof(RelaxCounters) <= sizeof(Counters), "counter block too small");
	RelaxCounters *d_rc = reinterpret_cast<RelaxCounters *>(priv->counters.p);
	RelaxCounters *h_rc = reinterpret_cast<RelaxCounters *>(priv->h_cnt);
	if (fresh) {
		if constexpr (sizeof(DT) == 4) hipLaunchKernelGGL(k_fill32, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, priv->dist.as<int32_t>(), (int64_t)cells, (int32_t)inf_label);
		else hipLaunchKernelGGL(k_fill64, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, priv->dist.as<int64_t>(), (int64_t)cells, inf_bits);
	}
	PGQ_HIP_TRY(hipMemsetAsync(priv->dirty[0].p, 0, (size_t)std::max<int64_t>(V, 1) * 8, st));
	PGQ_HIP_TRY(hipMemsetAsync(priv->dirty[1].p, 0, (size_t)std::max<int64_t>(V, 1) * 8, st));
	PGQ_HIP_TRY(hipMemsetAsync(priv->qflag.p, 0, (size_t)std::max<int64_t>(V, 1) * 4, st));
	PGQ_HIP_TRY(hipMemsetAsync(priv->tflag.p, 0, (size_t)std::max<int64_t>(V, 1) * 4, st));
	u32 epoch = 0, tepoch = 0;
	// the persistent grid of a round: exactly what is resident at once (each wavefront takes every nwaves-th vertex; a
	// grid larger than the chip makes the last workgroups start when the first finish: 8192 waves on 7168 slots was 2x)
	static std::mutex grid_lock;
	static unsigned grid_cached[3][64] = {}; // per label / weight type and device: a node's devices need not be alike
	unsigned grid;
	{
		std::lock_guard<std::mutex> g(grid_lock);
		int dev = 0;
		PGQ_HIP_TRY(hipGetDevice(&dev));
		unsigned &gc = grid_cached[type_tag - 1][dev & 63];
		if (!gc) {
			int per_cu = 0, cus = 0;
			PGQ_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_relax<T, DT>, 256, 0));
			PGQ_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
			gc = (unsigned)std::max(1, per_cu) * (unsigned)std::max(1, cus);
		}
		grid = gc;
	}
	for (int b = b0; b < nb; b += bstride) {
		const int64_t lo = bs[b], hi = bs[b + 1];
		if (lo == hi) continue;
		S.batches++;
		const int64_t base = (int64_t)b * LC;
		tepoch++;
		PGQ_HIP_TRY(hipMemsetAsync(d_rc, 0, sizeof(RelaxCounters), st));
		{
			KernelTimer kt(st, K_PREP);
			hipLaunchKernelGGL(k_cheapest_init<DT>, dim3(1), dim3(64), 0, st, ws->usrc.as<int32_t>(), (int64_t)U, base,
			                   priv->dist.as<DT>(), priv->dirty[0].as<u64>(), priv->tflag.as<u32>(), tepoch,
			                   priv->touched.as<int32_t>(), &d_rc->nq[0], &d_rc->tcount, priv->qbuf[0].as<int32_t>());
			kt.stop();
		}
		int par = 0;
		u32 nq_now = (u32)std::min<int64_t>(LC, (int64_t)U - base);
		const u32 small_limit = (u32)std::max(0, options().relax_small_limit);
		// band width of the ordered rounds: a fraction of the mean weight (0: plain Jacobi rounds, everything at once)
		T band = T(0);
		if (options().relax_delta_div > 0) {
			PGQ_TRY(ensure_weight_mean(c, ws));
			if constexpr (std::is_same<T, double>::value) band = c->w_mean / options().relax_delta_div;
			else band = (T)std::max<int64_t>(1, (int64_t)(c->w_mean / options().relax_delta_div));
		}
		auto bits_of = [](T x) -> long long {
			long long r;
			memcpy(&r, &x, 8);
			return r;
		};
		T thr_val = band;
		// "light edges first" (relax_light): the rounds run over weight-sorted lists under a cap on the edge weight that
		// doubles phase by phase — early phases touch a few per cent of the edges and give every lane a tight bound on
		// its destinations, later phases hardly relax anything (a vertex stops at the first edge whose weight cannot beat
		// its lanes' bounds) — until the cap reaches the largest bound (heavier edges cannot be on a cheaper path)
		// (a mean that is not a positive finite number — NaN / inf weights, an int64 sum that wrapped — gives no first cap:
		// plain rounds then)
		const bool heavy = options().relax_split != 0;
		static const bool trace = getenv("PGQ_RELAX_TRACE") != nullptr; // per-round line on stderr (measurement only)
		auto t_round = std::chrono::steady_clock::now();
		T wcap = T(0);
		if (light) {
			PGQ_TRY(ensure_weight_sorted(c, ws));
			PGQ_TRY(ensure_weight_mean(c, ws));
			if constexpr (std::is_same<T, double>::value) wcap = c->w_mean / std::max(1, options().relax_light_div);
			else wcap = (T)std::max<int64_t>(1, (int64_t)(c->w_mean / std::max(1, options().relax_light_div)));
		}
		const int32_t *r_adj = light ? c->wadj : c->adj;
		const T *r_w = light ? (const T *)c->wsorted : (const T *)c->w;
		for (;;) {
			if (!light && sizeof(DT) == 4) return fail(PGQ_ERR_HIP, "internal error: 4-byte labels outside the light-edges-first path");
			if (!light && nq_now <= small_limit) {
				// few changed vertices: rounds loop on the device inside one workgroup (8-byte labels only: labels_fit_32)
				const int max_rounds = 4096;
				KernelTimer kt(st, K_RELAX);
				hipLaunchKernelGGL(k_relax_small<T>, dim3(1), dim3(1024), 0, st, c->off, c->adj, (const T *)c->w,
				                   priv->dist.as<int64_t>(), priv->dirty[0].as<u64>(), priv->dirty[1].as<u64>(),
				                   priv->qbuf[0].as<int32_t>(), priv->qbuf[1].as<int32_t>(), d_rc, par, small_limit,
				                   priv->qflag.as<u32>(), epoch + 1, priv->tflag.as<u32>(), tepoch,
				                   priv->touched.as<int32_t>(), max_rounds);
				kt.stop();
				PGQ_HIP_TRY(hipMemcpyAsync(h_rc, d_rc, sizeof(RelaxCounters), hipMemcpyDeviceToHost, st));
				PGQ_HIP_TRY(hipStreamSynchronize(st));
				KernelTimer::flush();
				epoch += h_rc->rounds + 1;
				S.levels += h_rc->rounds;
				par ^= (int)(h_rc->rounds & 1u);
				nq_now = h_rc->nq[par];
				if (nq_now == 0) break;
				if (nq_now <= small_limit && h_rc->rounds > 0) continue; // hit max_rounds: go again
			}
			epoch++;
			// one launch instead of five fills per round (2459 rounds per 4096-pair step on the weighted knows graph)
			hipLaunchKernelGGL(k_round_reset, dim3(1), dim3(64), 0, st, d_rc, par ^ 1);
			const long long thr_bits = band > T(0) ? bits_of(thr_val) : (long long)0x7FFFFFFFFFFFFFFFll;
			{
				KernelTimer kt(st, K_RELAX);
				hipLaunchKernelGGL(k_lane_bounds<DT>, dim3((unsigned)std::min<int64_t>(blocks_for(hi - lo), 256)), dim3(256), 0, st, lo, hi,
				                   ws->skey.as<u32>(), ws->sdst.as<int32_t>(), (u32)base, priv->dist.as<DT>(), inf_bits, d_rc->bound);
				hipLaunchKernelGGL((k_relax<T, DT>), dim3(std::min(grid, std::max(1u, (nq_now + 3) / 4))), dim3(256), 0, st, c->off, r_adj, r_w,
				                   priv->dist.as<DT>(), priv->dirty[par].as<u64>(), priv->dirty[par ^ 1].as<u64>(),
				                   priv->qbuf[par].as<int32_t>(), &d_rc->nq[par], priv->qbuf[par ^ 1].as<int32_t>(),
				                   &d_rc->nq[par ^ 1], priv->tflag.as<u32>(), tepoch,
				                   priv->touched.as<int32_t>(), &d_rc->tcount, &d_rc->relaxed_edges, thr_bits,
				                   (const long long *)d_rc->bound, &d_rc->min_deferred, &d_rc->relaxed_vertices, light ? 1 : 0, wcap,
				                   0, heavy ? &d_rc->heavy : (u64 *)nullptr, priv->hv.as<int32_t>(), priv->hmask.as<u64>(),
				                   priv->hstart.as<u32>(), priv->hmap.as<u32>(), (const DT *)nullptr, (long long *)nullptr, (u32 *)nullptr,
				                   (const long long *)nullptr);
				if (heavy) // the long lists of the round, a chunk per wavefront
					hipLaunchKernelGGL((k_relax<T, DT>), dim3(grid), dim3(256), 0, st, c->off, r_adj, r_w,
					                   priv->dist.as<DT>(), priv->dirty[par].as<u64>(), priv->dirty[par ^ 1].as<u64>(),
					                   priv->qbuf[par].as<int32_t>(), &d_rc->nq[par], priv->qbuf[par ^ 1].as<int32_t>(),
					                   &d_rc->nq[par ^ 1], priv->tflag.as<u32>(), tepoch,
					                   priv->touched.as<int32_t>(), &d_rc->tcount, &d_rc->relaxed_edges, thr_bits,
					                   (const long long *)d_rc->bound, &d_rc->min_deferred, &d_rc->relaxed_vertices, light ? 1 : 0, wcap,
					                   1, &d_rc->heavy, priv->hv.as<int32_t>(), priv->hmask.as<u64>(), priv->hstart.as<u32>(),
					                   priv->hmap.as<u32>(), (const DT *)nullptr, (long long *)nullptr, (u32 *)nullptr, (const long long *)nullptr);
				kt.stop();
			}
			PGQ_HIP_TRY(hipMemcpyAsync(h_rc, d_rc, sizeof(RelaxCounters), hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
			KernelTimer::flush();
			if (trace) {
				const auto t1 = std::chrono::steady_clock::now();
				fprintf(stderr, "relax b=%d round=%lld cap=%g nq=%u expanded=%u heavy=%u/%u edges=%llu next=%u us=%.1f\n", b,
				        (long long)S.levels, (double)wcap, nq_now, h_rc->relaxed_vertices, (u32)(h_rc->heavy >> 32),
				        (u32)h_rc->heavy, (unsigned long long)h_rc->relaxed_edges, h_rc->nq[par ^ 1],
				        std::chrono::duration<double, std::micro>(t1 - t_round).count());
				t_round = t1;
			}
			S.levels++;
			par ^= 1;
			nq_now = h_rc->nq[par];
			if (nq_now == 0) {
				if (!light) break;
				// the phase has reached its fixpoint.  Done when no heavier edge can matter: the cap has reached the largest
				// bound of a lane (the bounds of the last round: labels only got smaller since) or the largest weight
				T max_bound = T(0);
				bool unbounded = false;
				for (int l = 0; l < LC; l++) {
					if (h_rc->bound[l] >= (long long)inf_bits) unbounded = true;
					T bv;
					memcpy(&bv, &h_rc->bound[l], 8);
					if (bv > max_bound) max_bound = bv;
				}
				// (negated comparisons: a cap that is NaN or inf ends the search like one that has reached the largest weight)
				T w_max;
				memcpy(&w_max, &c->w_max_bits, 8);
				if (!(wcap < w_max) || (!unbounded && !(wcap < max_bound))) break;
				if constexpr (std::is_same<T, double>::value) wcap = wcap + wcap;
				else wcap = wcap > std::numeric_limits<int64_t>::max() / 2 ? std::numeric_limits<int64_t>::max() : wcap + wcap;
				// every labelled vertex again, over the longer prefix of its list
				PGQ_HIP_TRY(hipMemsetAsync(&d_rc->nq[par], 0, 4, st));
				hipLaunchKernelGGL(k_redirty, dim3((unsigned)device_cus() * 4), dim3(256), 0, st, priv->touched.as<int32_t>(), &d_rc->tcount,
				                   priv->dirty[par].as<u64>(), priv->qbuf[par].as<int32_t>(), &d_rc->nq[par]);
				PGQ_HIP_TRY(hipMemcpyAsync(h_rc, d_rc, sizeof(RelaxCounters), hipMemcpyDeviceToHost, st));
				PGQ_HIP_TRY(hipStreamSynchronize(st));
				nq_now = h_rc->nq[par];
				if (nq_now == 0) break;
				continue;
			}
			if (band > T(0)) { // next band; an empty one is skipped: straight to the smallest label left
				thr_val = thr_val + band;
				if (h_rc->relaxed_vertices == 0 && h_rc->min_deferred != 0x7F7F7F7F7F7F7F7Fll) {
					T m;
					memcpy(&m, &h_rc->min_deferred, 8);
					if (m + band > thr_val) thr_val = m + band;
				}
			}
		}
		S.edges_scanned += (int64_t)h_rc->relaxed_edges;
		S.algo_bytes[K_RELAX] += (double)h_rc->relaxed_edges * (4.0 + 8.0 + 2.0 * (double)sizeof(DT) * LC);
		hipLaunchKernelGGL(k_cheapest_results<DT>, dim3(blocks_for(hi - lo)), dim3(256), 0, st, lo, hi, ws->skey.as<u32>(),
		                   ws->sidx.as<u32>(), ws->sdst.as<int32_t>(), (u32)base, priv->dist.as<DT>(), inf_bits,
		                   d_out, d_ok);
		hipLaunchKernelGGL(k_reset_touched<DT>, dim3((unsigned)device_cus() * 4), dim3(256), 0, st, priv->touched.as<int32_t>(), &d_rc->tcount,
		                   priv->dist.as<DT>(), inf_label);
	}
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	KernelTimer::flush();
	priv->dist_V = V; // every touched row is back at INF
	priv->dist_lanes = type_tag;
	return PGQ_OK;
}

// ---- general graphs, about one destination per source: both ends at once (relax_batches_bidir, round 6) ------------------
// The batched relaxation above gives a lane the distances from its source to EVERY vertex under the lane's bound: on the
// weighted knows graph (weights 1..999, mean degree 89) that is most of the graph per source, ~14 expansions per vertex and
// batch — 0.64 s per 4096 pairs, `k_relax` 94 % of it.  A single pair needs the ball of half the distance around each end:
// balls grow by e^(0.089 r) there, so two of radius D / 2 hold a few hundred settled vertices where one of radius D holds
// most of V.  Here a lane is one (src, dst) PAIR and a batch runs the same kernel from both ends — forward over the
// weight-sorted out-lists into `dist`, backward over the weight-sorted in-lists into `dist_b`:
//     mu[l]  = the best src -> dst length seen (the bound BOTH sides prune with; INF at first)
//     a round = k_relax forward, then k_relax backward, each expanding what is dirty with a label below the common cap C
//               (the kernel's threshold, read from the device) and below mu[l], labelling what gets under min(mu[l], 2 C) —
//               an expansion walks the few edges under 2 C - label, not its list; a vertex expanded on one side that carries
//               a label of the other side offers mu[l] = label + other label
//     a phase ends with a round in which neither side expanded anything (both at their fixpoint under C): a lane is finished
//               when mu[l] < 2 C, or when one of its sides has neither a labelled vertex left over the cap nor an edge the
//               cap cut (closure exhausted); then C rises — by a step, by a quarter, and past empty bands to the smallest
//               label left — and every labelled vertex is expanded again (k_redirty: the longer prefix of its list)
// Exactness (int64 sums, any order).  At a phase end every vertex with d(src, x) < min(C, mu) carries its exact forward
// label and has been expanded with it, likewise backward.  Let D < 2 C be the distance and x the last vertex of an optimal
// path with d(src, x) < C, y its successor: d(src, y) >= C, so d(y, dst) = D - d(src, y) < C, and both labels the two
// offer each other — d(src, x) + w forward on y, w + d(y, dst) backward on x — are at most D < 2 C: not cut.  The later of "x expanded
// forward for the last time" / "y expanded backward for the last time" reads the other's final label — x finds db[x] <=
// w + d(y, dst), or y finds df[y] <= d(src, x) + w — and offers D (launches of one batch never overlap, so one of the two
// IS later).  Hence mu < 2 C implies D <= mu < 2 C implies mu = D.  A side that has nothing labelled left below mu has
// expanded its whole closure below mu: were D < mu, dst (src) itself would have been expanded there and offered D.
// int64 weights only: a double's sum depends on the order of its additions (cheapest_path_length.cpp:29-36 folds from the
// source), so doubles keep the one-sided relaxation above.
// MEASURED (round 6, weighted knows graph, 512 pairs): 232 ms against 89 ms one-sided, so `relax_bidir` ships OFF.  That
// graph has no small balls: its hubs (1800 edges) sit within half a typical distance (D ~ 60..230, median 95) of nearly
// every vertex, a batch of 64 pairs relaxes 30 M edges from both ends against 40 M one-sided — and pays 83 rounds
// (9 phases of caps 15, 22, 29, 36, 45, 56, 70, 87, 108, each re-expanding what is labelled) instead of 35, every one of
// them a rescan of ~400 K labelled vertices per side.  Both variants move a 64-lane label row per relaxed edge for the
// one or two lanes that need it; that row traffic, not the number of edges, is what a round costs.  On
// graphs whose balls stay small (bounded degree, road-like) the two-ended search is the cheaper one; kept, tested, optional.
struct BiBlock {
	RelaxCounters rc[2]; // forward / backward: k_relax's round counters
	long long mu[64];    // per lane: best src -> dst length so far = both sides' bound; 0 once the lane is finished
	long long res[64];   // per lane: the answer (INF: no path)
	u32 alive[2][64];    // per side and lane: a labelled vertex was left unexpanded (over the cap) this round
	u32 done[64];
	long long cap, step;
	u32 active, phases;
};

template <typename DT>
__global__ __launch_bounds__(64) void k_bidir_init(int64_t lo, int nl, const int32_t *__restrict__ ssrc, const int32_t *__restrict__ sdst,
                                                   DT *__restrict__ dist_f, DT *__restrict__ dist_b, u64 *__restrict__ dirty_f,
                                                   u64 *__restrict__ dirty_b, u32 *__restrict__ tflag_f, u32 *__restrict__ tflag_b, u32 tepoch,
                                                   int32_t *__restrict__ touched_f, int32_t *__restrict__ touched_b,
                                                   int32_t *__restrict__ q_f, int32_t *__restrict__ q_b, BiBlock *__restrict__ bb,
                                                   long long cap0, long long step, long long inf) {
	const int t = threadIdx.x; // the block arrives zeroed
	bb->res[t] = inf;
	if (t < nl) {
		const int s = ssrc[lo + t], d = sdst[lo + t]; // s != d (trivial rows sort behind the lanes); several lanes may share an endpoint
		bb->mu[t] = inf;
		dist_f[(size_t)s * LC + t] = 0;
		if (atomicOr(&dirty_f[s], 1ull << t) == 0ull) q_f[atomicAdd(&bb->rc[0].nq[0], 1u)] = s;
		if (atomicExch(&tflag_f[s], tepoch) != tepoch) touched_f[atomicAdd(&bb->rc[0].tcount, 1u)] = s;
		dist_b[(size_t)d * LC + t] = 0;
		if (atomicOr(&dirty_b[d], 1ull << t) == 0ull) q_b[atomicAdd(&bb->rc[1].nq[0], 1u)] = d;
		if (atomicExch(&tflag_b[d], tepoch) != tepoch) touched_b[atomicAdd(&bb->rc[1].tcount, 1u)] = d;
	} else {
		bb->done[t] = 1; // no pair on this lane: bound 0, nothing is ever expanded for it
	}
	if (t == 0) {
		bb->cap = cap0;
		bb->step = step;
		bb->active = (u32)nl;
	}
}

__global__ __launch_bounds__(64) void k_bidir_round_reset(BiBlock *__restrict__ bb, int next_f, int next_b) {
	const int t = threadIdx.x;
	if (t < 2) {
		RelaxCounters *rc = &bb->rc[t];
		rc->nq[t == 0 ? next_f : next_b] = 0;
		rc->relaxed_vertices = 0;
		rc->min_deferred = 0x7F7F7F7F7F7F7F7Fll;
		rc->heavy = 0;
	}
}

// after the two sides' launches of a round (see the header above)
__global__ __launch_bounds__(64) void k_bidir_phase_end(BiBlock *__restrict__ bb) {
	const int t = threadIdx.x;
	const bool fix = bb->rc[0].relaxed_vertices == 0 && bb->rc[1].relaxed_vertices == 0;
	const long long C = bb->cap;
	if (fix && !bb->done[t]) {
		const long long m = bb->mu[t];
		if (m < 2 * C || !bb->alive[0][t] || !bb->alive[1][t]) {
			bb->res[t] = m;
			bb->mu[t] = 0; // the lane's bound: nothing of it is expanded or labelled any more
			bb->done[t] = 1;
		}
	}
	if (fix) { // the flags gather over a phase's rounds (an edge is cut in the round its vertex is expanded)
		bb->alive[0][t] = 0;
		bb->alive[1][t] = 0;
	}
	const u64 open = __ballot(!bb->done[t]);
	if (t == 0) {
		bb->active = (u32)__popcll(open);
		if (fix) {
			const long long md = min(bb->rc[0].min_deferred, bb->rc[1].min_deferred);
			long long nc = max(C + bb->step, C + C / 4);
			if (md != 0x7F7F7F7F7F7F7F7Fll && md + 1 > nc) nc = md + 1; // an empty band: straight to the smallest label left
			bb->cap = nc;
			bb->phases++;
		}
	}
}

__global__ __launch_bounds__(64) void k_bidir_results(int64_t lo, int nl, const u32 *__restrict__ sidx, const BiBlock *__restrict__ bb,
                                                      long long inf, int64_t *__restrict__ out, uint8_t *__restrict__ ok) {
	const int t = threadIdx.x;
	if (t >= nl) return;
	const u32 row = sidx[lo + t];
	const long long d = bb->res[t];
	ok[row] = d == inf ? 0 : 1;
	out[row] = d == inf ? 0 : d;
}

static std::mutex g_rws_lock;
// the in-lists sorted by weight (radj / rw -> rwadj / rwsorted), once per CSR: ensure_weight_sorted for the other direction
static int ensure_reverse_sorted(pgq_csr *c, Workspace *ws) {
	PGQ_TRY(ensure_reverse_weights(c, ws));
	std::lock_guard<std::mutex> g(g_rws_lock);
	if (c->rwadj || c->E == 0 || !c->rw) return PGQ_OK;
	hipStream_t st = ws->stream;
	const int64_t E = c->E;
	DevBuf tmp;
	int32_t *rwadj = nullptr;
	void *rwsorted = nullptr;
	auto body = [&]() -> int {
		PGQ_TRY(dev_alloc_as(&rwadj, (size_t)E + 4));
		PGQ_TRY(dev_alloc(&rwsorted, (size_t)E * 8));
		size_t sb = 0;
		const unsigned long long *keys = (const unsigned long long *)c->rw; // int64 weights >= 0: the bit pattern is the value
		PGQ_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, sb, keys, (unsigned long long *)rwsorted, c->radj, rwadj, (int)E, (int)c->V,
		                                                        c->roff, c->roff + 1, 0, 64, st));
		PGQ_TRY(tmp.reserve(sb + 16));
		PGQ_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp.p, sb, keys, (unsigned long long *)rwsorted, c->radj, rwadj, (int)E, (int)c->V,
		                                                        c->roff, c->roff + 1, 0, 64, st));
		PGQ_HIP_TRY(hipStreamSynchronize(st));
		return PGQ_OK;
	};
	const int rc = body();
	(void)hipStreamSynchronize(st);
	tmp.release();
	if (rc != PGQ_OK) {
		dev_free(rwadj);
		dev_free(rwsorted);
		return rc;
	}
	c->rwsorted = rwsorted;
	c->rwadj = rwadj;
	return PGQ_OK;
}

// The pair batches b0, b0 + bstride, ... < nb of a call: batch b = the sorted rows [64 b, 64 b + 64) of the R rows that need a
// search (`ws`: sorted rows, read-only; `priv`: everything a batch writes, and the stream — like relax_batches).
template <typename DT>
static int relax_batches_bidir(pgq_csr *c, Workspace *ws, Workspace *priv, int b0, int bstride, int nb, int64_t R, int64_t *d_out,
                               uint8_t *d_ok) {
	using T = int64_t;
	hipStream_t st = priv->stream;
	const int64_t V = std::max<int64_t>(c->V, 1);
	const int64_t inf_bits = Inf<T>::bits;
	const DT inf_label = (DT)Lab<DT>::unlabelled(inf_bits);
	pgq_stats_t &S = tstats().s;
	const size_t cells = (size_t)V * LC;
	const int type_tag = sizeof(DT) == 4 ? 3 : 1;
	DevBuf *dist[2] = { &priv->dist, &priv->dist_b };
	DevBuf *dirty[2][2] = { { &priv->dirty[0], &priv->dirty[1] }, { &priv->dirty_b[0], &priv->dirty_b[1] } };
	DevBuf *qbuf[2][2] = { { &priv->qbuf[0], &priv->qbuf[1] }, { &priv->qbuf_b[0], &priv->qbuf_b[1] } };
	DevBuf *touched[2] = { &priv->touched, &priv->touched_b }, *tflag[2] = { &priv->tflag, &priv->tflag_b };
	int64_t *dist_V[2] = { &priv->dist_V, &priv->dist_b_V };
	int *dist_tag[2] = { &priv->dist_lanes, &priv->dist_b_lanes };
	for (int s = 0; s < 2; s++) {
		const bool fresh = dist[s]->cap < cells * sizeof(DT) || *dist_V[s] != c->V || *dist_tag[s] != type_tag;
		*dist_V[s] = -1; // stays invalid if we bail out half-way; restored at the end
		PGQ_TRY(dist[s]->reserve(cells * sizeof(DT)));
		for (int k = 0; k < 2; k++) {
			PGQ_TRY(dirty[s][k]->reserve((size_t)V * 8));
			PGQ_TRY(qbuf[s][k]->reserve((size_t)V * 4));
		}
		PGQ_TRY(touched[s]->reserve((size_t)V * 4));
		PGQ_TRY(tflag[s]->reserve((size_t)V * 4));
		if (fresh) {
			if constexpr (sizeof(DT) == 4) hipLaunchKernelGGL(k_fill32, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, dist[s]->template as<int32_t>(), (int64_t)cells, (int32_t)inf_label);
			else hipLaunchKernelGGL(k_fill64, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, dist[s]->template as<int64_t>(), (int64_t)cells, inf_bits);
		}
		for (int k = 0; k < 2; k++) PGQ_HIP_TRY(hipMemsetAsync(dirty[s][k]->p, 0, (size_t)V * 8, st));
		PGQ_HIP_TRY(hipMemsetAsync(tflag[s]->p, 0, (size_t)V * 4, st));
	}
	PGQ_TRY(priv->hv.reserve((size_t)V * 4));
	PGQ_TRY(priv->hmask.reserve((size_t)V * 8));
	PGQ_TRY(priv->hstart.reserve((size_t)V * 4));
	PGQ_TRY(priv->hmap.reserve((size_t)(c->E / 64 + V) * 4));
	PGQ_TRY(priv->bi_block.reserve(sizeof(BiBlock)));
	if (!priv->h_bi) PGQ_HIP_TRY(hipHostMalloc(&priv->h_bi, sizeof(BiBlock)));
	BiBlock *bb = priv->bi_block.as<BiBlock>();
	const BiBlock *hb = static_cast<const BiBlock *>(priv->h_bi);
	unsigned grid;
	{
		static std::mutex grid_lock;
		static unsigned grid_cached[2][64] = {};
		std::lock_guard<std::mutex> g(grid_lock);
		int dev = 0;
		PGQ_HIP_TRY(hipGetDevice(&dev));
		unsigned &gc = grid_cached[sizeof(DT) == 4 ? 0 : 1][dev & 63];
		if (!gc) {
			int per_cu = 0, cus = 0;
			PGQ_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_relax<T, DT>, 256, 0));
			PGQ_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
			gc = (unsigned)std::max(1, per_cu) * (unsigned)std::max(1, cus);
		}
		grid = gc;
	}
	const Options &opt = options();
	// (the mean of a shortest path's edges is far below the mean weight: the caps are fractions of twice the mean)
	const long long cap0 = std::max<long long>(1, (long long)(2.0 * c->w_mean / std::max(1, opt.relax_bidir_c0_div)));
	const long long step = std::max<long long>(1, (long long)(2.0 * c->w_mean / std::max(1, opt.relax_bidir_step_div)));
	const int64_t *xoff[2] = { c->off, c->roff };
	const int32_t *xadj[2] = { c->wadj, c->rwadj };
	const T *xw[2] = { (const T *)c->wsorted, (const T *)c->rwsorted };
	static const bool trace = getenv("PGQ_RELAX_TRACE") != nullptr;
	u32 tepoch = 0;
	for (int b = b0; b < nb; b += bstride) {
		const int64_t lo = (int64_t)b * LC;
		const int nl = (int)std::min<int64_t>(LC, R - lo);
		if (nl <= 0) continue;
		S.batches++;
		tepoch++;
		PGQ_HIP_TRY(hipMemsetAsync(bb, 0, sizeof(BiBlock), st));
		{
			KernelTimer kt(st, K_PREP);
			hipLaunchKernelGGL(k_bidir_init<DT>, dim3(1), dim3(64), 0, st, lo, nl, ws->ssrc.as<int32_t>(), ws->sdst.as<int32_t>(), dist[0]->template as<DT>(),
			                   dist[1]->template as<DT>(), dirty[0][0]->template as<u64>(), dirty[1][0]->template as<u64>(), tflag[0]->template as<u32>(),
			                   tflag[1]->template as<u32>(), tepoch, touched[0]->template as<int32_t>(), touched[1]->template as<int32_t>(),
			                   qbuf[0][0]->template as<int32_t>(), qbuf[1][0]->template as<int32_t>(), bb, cap0, step, (long long)inf_bits);
			kt.stop();
		}
		int par[2] = { 0, 0 };
		u32 nq[2] = { (u32)nl, (u32)nl }, phases_seen = 0;
		for (int64_t round = 0;; round++) {
			if (round > (int64_t)1 << 22) return fail(PGQ_ERR_HIP, "internal error: the bidirectional relaxation does not terminate");
			hipLaunchKernelGGL(k_bidir_round_reset, dim3(1), dim3(64), 0, st, bb, par[0] ^ 1, par[1] ^ 1);
			{
				KernelTimer kt(st, K_RELAX);
				for (int s = 0; s < 2; s++) {
					if (nq[s] == 0) continue;
					RelaxCounters *rc = &bb->rc[s];
					for (int hp = 0; hp < 2; hp++)
						hipLaunchKernelGGL((k_relax<T, DT>), dim3(hp ? grid : std::min(grid, std::max(1u, (nq[s] + 3) / 4))), dim3(256), 0, st, xoff[s], xadj[s], xw[s],
						                   dist[s]->template as<DT>(), dirty[s][par[s]]->template as<u64>(), dirty[s][par[s] ^ 1]->template as<u64>(),
						                   qbuf[s][par[s]]->template as<int32_t>(), &rc->nq[par[s]], qbuf[s][par[s] ^ 1]->template as<int32_t>(),
						                   &rc->nq[par[s] ^ 1], tflag[s]->template as<u32>(), tepoch, touched[s]->template as<int32_t>(), &rc->tcount,
						                   &rc->relaxed_edges, (long long)0, (const long long *)bb->mu, &rc->min_deferred, &rc->relaxed_vertices, 1,
						                   std::numeric_limits<T>::max(), hp, &rc->heavy, priv->hv.as<int32_t>(), priv->hmask.as<u64>(),
						                   priv->hstart.as<u32>(), priv->hmap.as<u32>(), (const DT *)dist[s ^ 1]->template as<DT>(), bb->mu, bb->alive[s],
						                   (const long long *)&bb->cap);
				}
				hipLaunchKernelGGL(k_bidir_phase_end, dim3(1), dim3(64), 0, st, bb);
				kt.stop();
			}
			PGQ_HIP_TRY(hipMemcpyAsync(priv->h_bi, bb, sizeof(BiBlock), hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
			KernelTimer::flush();
			S.levels++;
			if (trace)
				fprintf(stderr, "bidir b=%d round=%lld cap=%lld phases=%u nq=%u/%u expanded=%u/%u next=%u/%u active=%u edges=%llu/%llu\n", b, (long long)round,
				        hb->cap, hb->phases, nq[0], nq[1], hb->rc[0].relaxed_vertices, hb->rc[1].relaxed_vertices, hb->rc[0].nq[par[0] ^ 1],
				        hb->rc[1].nq[par[1] ^ 1], hb->active, (unsigned long long)hb->rc[0].relaxed_edges, (unsigned long long)hb->rc[1].relaxed_edges);
			for (int s = 0; s < 2; s++) {
				if (nq[s] == 0) continue; // (nothing was launched: the side's queues and parity stay as they are)
				par[s] ^= 1;
				nq[s] = hb->rc[s].nq[par[s]];
			}
			if (hb->active == 0) break;
			if (hb->phases != phases_seen) { // a higher cap: every labelled vertex again, over the longer prefix of its list
				phases_seen = hb->phases;
				for (int s = 0; s < 2; s++) {
					hipLaunchKernelGGL(k_redirty, dim3((unsigned)device_cus() * 4), dim3(256), 0, st, touched[s]->template as<int32_t>(), &bb->rc[s].tcount,
					                   dirty[s][par[s]]->template as<u64>(), qbuf[s][par[s]]->template as<int32_t>(), &bb->rc[s].nq[par[s]]);
					nq[s] = hb->rc[s].tcount;
				}
			}
		}
		const u64 edges = hb->rc[0].relaxed_edges + hb->rc[1].relaxed_edges;
		S.edges_scanned += (int64_t)edges;
		S.algo_bytes[K_RELAX] += (double)edges * (4.0 + 8.0 + 2.0 * (double)sizeof(DT) * LC);
		hipLaunchKernelGGL(k_bidir_results, dim3(1), dim3(64), 0, st, lo, nl, ws->sidx.as<u32>(), bb, (long long)inf_bits, d_out, d_ok);
		for (int s = 0; s < 2; s++)
			hipLaunchKernelGGL(k_reset_touched<DT>, dim3((unsigned)device_cus() * 4), dim3(256), 0, st, touched[s]->template as<int32_t>(), &bb->rc[s].tcount,
			                   dist[s]->template as<DT>(), inf_label);
		// what the finished lanes left dirty: the next batch starts from clean words
		for (int s = 0; s < 2; s++)
			for (int k = 0; k < 2; k++) PGQ_HIP_TRY(hipMemsetAsync(dirty[s][k]->p, 0, (size_t)V * 8, st));
	}
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	KernelTimer::flush();
	for (int s = 0; s < 2; s++) {
		*dist_V[s] = c->V; // every touched row is back at INF
		*dist_tag[s] = type_tag;
	}
	return PGQ_OK;
}

template <typename T>
static int cheapest_device(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                           int64_t *d_out, uint8_t *d_ok, bool chain) {
	if (chain && options().chain && n > 0 && n < (1LL << 31)) return cheapest_with_chains<T>(c, ws, n, d_src, d_dst, d_out, d_ok);
	hipStream_t st = ws->stream;
	pgq_stats_t &S = tstats().s;
	S.pairs += n;
	if (n == 0) return PGQ_OK;
	if (n >= (1LL << 31)) return fail(PGQ_ERR_INVALID_ARG, "more than 2^31-1 rows in one call");
	u32 U = 0;
	PGQ_TRY(prepare_lanes(c, ws, n, d_src, d_dst, &U));
	S.unique_sources += U;
	const int nb = (int)(((int64_t)U + LC - 1) / LC);
	PGQ_TRY(batch_bounds(ws, n, LC, nb));
	const int64_t *bs = ws->h_bstart;
	if (light_edges_first(c)) { // once per CSR, before the workers need them
		PGQ_TRY(ensure_weight_sorted(c, ws));
		PGQ_TRY(ensure_weight_mean(c, ws));
	} else if (options().relax_delta_div > 0) {
		PGQ_TRY(ensure_weight_mean(c, ws));
	}
	// Independent batches overlap on several streams (one host thread each, like the lane batches of the unweighted
	// search): the rounds with few changed vertices — a third of a batch's rounds — leave most of the chip idle.
	int workers = std::max(1, std::min(options().relax_streams > 0 ? options().relax_streams : options().streams, nb));
	if (workers > 1) { // every extra worker holds its own label array (V x 64 x 8 bytes): only while half the free memory covers them
		size_t free_b = 0, total_b = 0;
		PGQ_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		const size_t per_worker = (size_t)std::max<int64_t>(c->V, 1) * (LC * 8 + 64) + (size_t)(c->E / 64) * 4;
		workers = (int)std::min<size_t>((size_t)workers, 1 + free_b / 2 / per_worker);
	}
	// decided once for the whole call, handed to every worker (the weight-sorted copy above has left the mean and the largest
	// weight on the handle; a mean that is not a positive finite number gives no first cap: plain rounds)
	const bool light = light_edges_first(c) && c->wadj && c->w_mean > 0 && c->w_mean < 1e300;
	const bool narrow = light && labels_fit_32<T>(c);
	// about one destination per source (a list of pairs, not a cross product): every row is a lane of its own, searched from
	// both ends (relax_batches_bidir); many destinations per source share their source's lane as before
	const int64_t R = bs[nb]; // the rows that need a search (sorted by source; trivial and NULL rows behind them)
	bool bidir = false;
	if constexpr (std::is_same<T, int64_t>::value)
		bidir = options().relax_bidir && light && R > 0 && R <= (int64_t)std::max(1, options().relax_bidir_rows) * (int64_t)U;
	int nb_run = nb;
	if (bidir) {
		PGQ_TRY(ensure_reverse_sorted(c, ws));
		bidir = c->rwadj != nullptr;
		if (bidir) {
			nb_run = (int)((R + LC - 1) / LC);
			workers = std::max(1, std::min(workers, nb_run));
		}
	}
	auto run_relax = [&](Workspace *priv, int b0, int bstride) -> int {
		if constexpr (std::is_same<T, int64_t>::value) {
			if (bidir) return narrow ? relax_batches_bidir<int32_t>(c, ws, priv, b0, bstride, nb_run, R, d_out, d_ok)
			                         : relax_batches_bidir<int64_t>(c, ws, priv, b0, bstride, nb_run, R, d_out, d_ok);
			if (narrow) return relax_batches<T, int32_t>(c, ws, priv, b0, bstride, nb, U, d_out, d_ok, light);
		}
		return relax_batches<T, int64_t>(c, ws, priv, b0, bstride, nb, U, d_out, d_ok, light);
	};
	int rc = PGQ_OK;
	if (workers == 1) {
		rc = run_relax(ws, 0, 1);
	} else {
		std::vector<WorkspaceLease> leases((size_t)workers - 1);
		for (auto &l : leases) PGQ_TRY(l.acquire());
		std::vector<int> rcs((size_t)workers, PGQ_OK);
		std::vector<std::string> errs((size_t)workers);
		std::vector<pgq_stats_t> wstats((size_t)workers);
		std::vector<std::shared_ptr<WorkerTask>> pool;
		const int dev = current_device();
		Options *const parent_opt = options_override();
		for (int t = 1; t < workers; t++)
			pool.push_back(worker_submit(dev, [&, t]() {
				OptionScope opt_scope(parent_opt);
				bind_thread_device(dev);
				int r = ensure_init();
				if (r == PGQ_OK) {
					(void)pgq_reset_stats();
					r = run_relax(leases[(size_t)t - 1].ws, t, workers);
				}
				rcs[(size_t)t] = r;
				if (r != PGQ_OK) errs[(size_t)t] = pgq_last_error();
				wstats[(size_t)t] = tstats().s;
			}));
		rcs[0] = run_relax(ws, 0, workers);
		for (size_t k = 0; k < pool.size(); k++) { // a job that threw never wrote its return code: take the pool's word for it
			const int wr = worker_wait(pool[k]);
			if (wr != PGQ_OK) {
				rcs[k + 1] = wr;
				errs[k + 1] = pgq_last_error();
			}
		}
		for (int t = 0; t < workers; t++) {
			if (rcs[(size_t)t] != PGQ_OK && rc == PGQ_OK) {
				rc = rcs[(size_t)t];
				if (t > 0) set_error(errs[(size_t)t]);
			}
			if (t > 0) merge_stats(S, wstats[(size_t)t]);
		}
	}
	if (rc != PGQ_OK) return rc;
	const int64_t lo_t = bs[nb + 1], lo_n = bs[nb + 2];
	if (n > lo_t)
		hipLaunchKernelGGL(k_cheapest_tails, dim3(blocks_for(n - lo_t)), dim3(256), 0, st, lo_t, lo_n, n,
		                   ws->sidx.as<u32>(), d_out, d_ok);
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	KernelTimer::flush();
	return PGQ_OK;
}

static int check_weighted(pgq_csr_t *csr) {
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (csr->w_type == PGQ_W_NONE || !csr->w)
		return fail(PGQ_ERR_NOT_WEIGHTED, "Constraint Error: Need to initialize CSR before doing cheapest path");
	if (csr->has_negative_weight)
		return fail(PGQ_ERR_UNSUPPORTED, "cheapest_path_length: negative edge weights are not supported");
	return PGQ_OK;
}

} // namespace pgq

using namespace pgq;

extern "C" {

int pgq_cheapest_path_length_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                         void *d_out, uint8_t *d_out_valid) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	PGQ_TRY(check_weighted(csr));
	if (n < 0 || (n > 0 && (!d_src || !d_dst || !d_out || !d_out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL device array");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	if (csr->w_type == PGQ_W_INT64)
		return cheapest_device<int64_t>(csr, lease.ws, n, d_src, d_dst, (int64_t *)d_out, d_out_valid, true);
	return cheapest_device<double>(csr, lease.ws, n, d_src, d_dst, (int64_t *)d_out, d_out_valid, true);
}

int pgq_cheapest_path_length(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, void *out,
                             uint64_t *out_valid) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	PGQ_TRY(check_weighted(csr));
	if (V != csr->V) return fail(PGQ_ERR_INVALID_ARG, "V does not match the uploaded CSR");
	if (n < 0 || (n > 0 && (!out || !out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL output");
	if (n == 0) return PGQ_OK;
	FlatPairs fp;
	PGQ_TRY(flatten_pairs(V, n, src, dst, fp, true));
	for (int64_t i = 0; i < n; i++)
		if (!fp.dst_valid[i]) fp.src[i] = -1; // NULL dst -> NULL (cheapest_path_length.cpp:74-76)
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *ws = lease.ws;
	PGQ_TRY(ws->in_src.reserve((size_t)n * 8));
	PGQ_TRY(ws->in_dst.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_val.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_ok.reserve((size_t)n));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, fp.src.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_dst.p, fp.dst.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	int rc;
	if (csr->w_type == PGQ_W_INT64)
		rc = cheapest_device<int64_t>(csr, ws, n, ws->in_src.as<int64_t>(), ws->in_dst.as<int64_t>(),
		                              ws->out_val.as<int64_t>(), ws->out_ok.as<uint8_t>(), true);
	else
		rc = cheapest_device<double>(csr, ws, n, ws->in_src.as<int64_t>(), ws->in_dst.as<int64_t>(),
		                             ws->out_val.as<int64_t>(), ws->out_ok.as<uint8_t>(), true);
	PGQ_TRY(rc);
	std::vector<uint8_t> ok(n);
	PGQ_TRY(staged_download(out, ws->out_val.p, (size_t)n * 8, ws->stream));
	PGQ_TRY(staged_download(ok.data(), ws->out_ok.p, (size_t)n, ws->stream));
	mask_fill_valid(out_valid, n);
	for (int64_t i = 0; i < n; i++)
		if (!ok[i]) mask_set_invalid(out_valid, i);
	return PGQ_OK;
}

} // extern "C"

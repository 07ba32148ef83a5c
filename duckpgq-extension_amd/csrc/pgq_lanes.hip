// pgq_lanes.hip — bottom-up BFS level for sparse frontiers, "lane-list" form (k_compact_lanes + k_pull_lanes).
//
// Same recurrence as k_pull / iterativelength.cpp:18-30 evaluated per destination:
//     next[n] = (OR over in-neighbours v of frontier[v]) & active & ~seen[n];   seen[n] |= next[n]
// organised for the regime MS-BFS spends most of its time in on social graphs (level 2 of a 2048-source batch on the
// SF100 knows graph: 49 % of the 39.9 M in-edges leave a frontier vertex, a frontier vertex carries 1.4 lanes on
// average, the fullest 18): there a frontier vertex is better described by the LIST of the lanes it carries than by
// lane-words.
//
//   k_compact_lanes   packs the frontier: per 64-vertex block {bit map, index of its first record}, and one dense
//                     16-byte record per frontier vertex holding up to ten 12-bit lane ids (WD <= 2: the raw
//                     lane-words); a vertex with more lanes is flagged "overflow" and served from its dense row.
//   k_pull_lanes      a wavefront owns a work part (<= 16 consecutive vertices, no hubs, built at upload) and walks
//                     the part's contiguous in-adjacency one entry per lane, UN x 64 entries in flight.  The entry
//                     word carries the in-neighbour and the owner row (rpk = radj | owner << 28: one coalesced
//                     4-byte load per in-edge).  A lane tests the neighbour's bit (bit map in LDS when it fits),
//                     fetches the neighbour's 16-byte record (one L2 request, no dependent second fetch) and sets
//                     one accumulator bit per lane id with ds_or_b32.  No wanted-word masks, no packed-word
//                     offsets, no spill queue: `& active & ~seen` is applied once per row in the epilogue.
//
// HBM-side algorithmic bytes per launch: 4*S (S in-edges scanned) + 16*H (H in-edges leaving a frontier vertex: one
// record each) + V*(4 + 24*WD) (seen read, next written, seen written, non-empty-word mask).
#include <algorithm>
#include <atomic>

#include "pgq_search.h"

namespace pgq {

struct __attribute__((aligned(16))) BlkInfo {
	u64 bits; // frontier vertices of the 64-vertex block
	u32 base; // record index of the block's first frontier vertex
	u32 pad;
};

constexpr int kLaneFields = 10;     // 12-bit lane ids per record
constexpr int kGroupBlocks = 8;     // 64-vertex blocks per group (one k_compact_lanes wavefront writes whole groups)
constexpr u32 kOverflowCount = 255; // record count byte: lanes do not fit, read the dense row instead

// ---- frontier packing -----------------------------------------------------------------------------------------------
// Every wavefront owns a contiguous range of 64-vertex blocks: pass 1 counts its frontier vertices, one atomicAdd
// claims its slice of recs[], pass 2 writes block infos and records (record of vertex v sits at
// blk[v/64].base + popcount(bits of the block below v)).
template <int WD>
__global__ __launch_bounds__(256) void k_compact_lanes(const u32 *__restrict__ nz, const u64 *__restrict__ front, int64_t V,
                                                       BlkInfo *__restrict__ blk, uint4 *__restrict__ recs,
                                                       u32 *__restrict__ totals, int stop_limit,
                                                       const Counters *__restrict__ cnt) {
	if (level_is_off(cnt, stop_limit)) return;
	const int lane = threadIdx.x & 63;
	const int64_t wave = (int64_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int64_t nwaves = (int64_t)((gridDim.x * blockDim.x) >> 6);
	// ranges are whole groups of kGroupBlocks blocks: k_pull_lanes keeps 16-bit bases relative to the group's first block
	const int64_t per = ((V + nwaves - 1) / nwaves + 64 * kGroupBlocks - 1) / (64 * kGroupBlocks) * (64 * kGroupBlocks);
	const int64_t v0 = wave * per, v1 = min(v0 + per, (V + 63) & ~63ll);
	u32 myv = 0;
	for (int64_t v = v0 + lane; v < v1; v += 64) myv += (v < V && nz[v] != 0) ? 1u : 0u;
	for (int o = 32; o > 0; o >>= 1) myv += __shfl_xor(myv, o);
	u32 vrun = 0;
	if (myv) {
		if (lane == 0) vrun = atomicAdd(&totals[0], myv);
		vrun = __shfl(vrun, 0);
	}
	for (int64_t vb = v0; vb < v1; vb += 64) {
		const int64_t v = vb + lane;
		const u32 m = v < V ? nz[v] : 0u;
		const u64 any = __ballot(m != 0);
		if (lane == 0) {
			BlkInfo bi;
			bi.bits = any;
			bi.base = vrun;
			bi.pad = 0;
			blk[vb >> 6] = bi;
		}
		if (!any) continue;
		if (m) {
			const u32 slot = vrun + (u32)__popcll(any & ((1ull << lane) - 1ull));
			uint4 rec = make_uint4(0, 0, 0, 0);
			if constexpr (WD <= 2) {
				const u64 w0 = front[(size_t)v * WD];
				const u64 w1 = WD == 2 ? front[(size_t)v * WD + 1] : 0ull;
				rec = make_uint4((u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32));
			} else {
				// w: ids 2..9 packed 12 bits each (96 bits); x: ids 0 and 1 with the count byte on top
				unsigned __int128 r = 0;
				u32 first = 0;
				u32 count = 0;
				bool ovf = __popc(m) > kLaneFields;
				u32 rest = m;
				while (rest && !ovf) {
					const int w = __ffs((int)rest) - 1;
					rest &= rest - 1;
					u64 x = front[(size_t)v * WD + w];
					if (count + (u32)__popcll(x) > (u32)kLaneFields) {
						ovf = true;
						break;
					}
					while (x) {
						const int b = __ffsll((long long)x) - 1;
						x &= x - 1;
						const u32 id = (u32)(w * 64 + b);
						if (count < 2) first |= id << (12 * count);
						else r |= (unsigned __int128)id << (12 * (count - 2));
						count++;
					}
				}
				const u32 cb = ovf ? kOverflowCount : count;
				rec = make_uint4((u32)r, (u32)(r >> 32), (u32)(r >> 64), first | (cb << 24));
			}
			recs[slot] = rec;
		}
		vrun += (u32)__popcll(any);
	}
}

// ---- the level kernel ----------------------------------------------------------------------------------------------
// lane id J (2..9) of a record's tail words (ids 2..9 packed 12 bits each in three words)
template <int J> __device__ __forceinline__ u32 tail_field(u32 t0, u32 t1, u32 t2) {
	constexpr int bit = 12 * (J - 2), w = bit >> 5, sh = bit & 31;
	const u32 a = w == 0 ? t0 : (w == 1 ? t1 : t2);
	if constexpr (sh <= 20) {
		return (a >> sh) & 0xFFFu;
	} else {
		const u32 b = w == 0 ? t1 : t2;
		return __builtin_amdgcn_alignbit(b, a, sh) & 0xFFFu;
	}
}

__device__ __forceinline__ void set_lane_bit(u32 *acc32, u32 ownbase, u32 id) {
	atomicOr(&acc32[ownbase + (id >> 5)], 1u << (id & 31u));
}

template <int J> struct TailLoop {
	// sets accumulator bit (owner row, lane id J) for the lanes whose record holds more than J ids; stops at the
	// first J no lane of the wavefront needs
	static __device__ __forceinline__ void run(u32 *acc32, u32 ownbase, u32 t0, u32 t1, u32 t2, u32 count) {
		if (!__any(count > (u32)J)) return;
		if (count > (u32)J) set_lane_bit(acc32, ownbase, tail_field<J>(t0, t1, t2));
		if constexpr (J + 1 < kLaneFields) TailLoop<J + 1>::run(acc32, ownbase, t0, t1, t2, count);
	}
};

// VEC lane-words per lane in the row prologue/epilogue (16-byte accesses when a row holds at least two words)
template <int VEC> struct RowVec;
template <> struct RowVec<1> {
	typedef u64 type;
	static __device__ __forceinline__ u64 zero() { return 0; }
	static __device__ __forceinline__ u32 fold(u64 a, u64 act, u64 s, u64 &fresh, u64 &upd) {
		fresh = a & act & ~s;
		upd = s | fresh;
		return fresh ? 1u : 0u;
	}
};
template <> struct RowVec<2> {
	typedef ulonglong2 type;
	static __device__ __forceinline__ ulonglong2 zero() { return make_ulonglong2(0, 0); }
	static __device__ __forceinline__ u32 fold(const ulonglong2 &a, const ulonglong2 &act, const ulonglong2 &s,
	                                           ulonglong2 &fresh, ulonglong2 &upd) {
		fresh.x = a.x & act.x & ~s.x;
		fresh.y = a.y & act.y & ~s.y;
		upd.x = s.x | fresh.x;
		upd.y = s.y | fresh.y;
		return (fresh.x ? 1u : 0u) | (fresh.y ? 2u : 0u);
	}
};

template <int WD, int UN, bool LDSMAP>
__global__ __launch_bounds__(LDSMAP ? 1024 : 256) void k_pull_lanes(
    const int64_t *__restrict__ roff, const u32 *__restrict__ rpk, const int64_t *__restrict__ off,
    const int32_t *__restrict__ parts, int n_parts, const BlkInfo *__restrict__ blk, const uint4 *__restrict__ recs,
    const u64 *__restrict__ front, u64 *__restrict__ seen, u64 *__restrict__ next, u32 *__restrict__ nz_next,
    const u64 *__restrict__ active, int n_blk, u32 rec_bytes, int stop_limit, Counters *__restrict__ cnt) {
	constexpr int WPB = LDSMAP ? 16 : 4;
	constexpr int NV = 16; // vertices per part (host-built parts never hold more; the owner row is 4 bits of rpk)
	constexpr int VEC = WD >= 2 ? 2 : 1;
	constexpr int RPL = (NV * WD / VEC + 63) / 64; // row vectors per lane
	constexpr int QCAP = 64; // tail queue entries per wavefront
	__shared__ __attribute__((aligned(16))) u64 s_acc[WPB][NV * WD];
	__shared__ __attribute__((aligned(16))) uint4 s_queue[WPB][QCAP];
	__shared__ u32 s_nzn[WPB][NV];
	__shared__ u64 red[WPB][5];
	// LDSMAP: n_blk bit-map words, n_blk 16-bit record bases relative to their group of kGroupBlocks blocks (records of
	// a group are contiguous: one k_compact_lanes wavefront wrote them), one 32-bit base per group
	extern __shared__ u64 s_dyn[];
	if (level_is_off(cnt, stop_limit)) return;
	u64 *s_bits = s_dyn;
	unsigned short *s_base = reinterpret_cast<unsigned short *>(s_dyn + n_blk);
	u32 *s_super = reinterpret_cast<u32 *>(s_base + ((n_blk + 1) & ~1));
	if constexpr (LDSMAP) {
		for (int i = threadIdx.x; i < n_blk; i += WPB * 64) {
			const BlkInfo bi = blk[i];
			const u32 sb = blk[i & ~(kGroupBlocks - 1)].base;
			s_bits[i] = bi.bits;
			s_base[i] = (unsigned short)(bi.base - sb);
			if ((i & (kGroupBlocks - 1)) == 0) s_super[i / kGroupBlocks] = sb;
		}
		__syncthreads();
	}
	// records are fetched through a buffer descriptor: a lane whose in-neighbour is not in the frontier presents an
	// out-of-range offset, which returns zeros without a memory request (no divergent branch around the load)
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)recs, 0, (int)rec_bytes, 0x00020000);
	const int lane = threadIdx.x & 63;
	const int wib = threadIdx.x >> 6;
	u64 *acc = s_acc[wib];
	u32 *acc32 = reinterpret_cast<u32 *>(acc);
	u32 *nzn = s_nzn[wib];
	uint4 *queue = s_queue[wib];
	const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (gridDim.x * blockDim.x) >> 6;
	u64 nf = 0, mf = 0, scanned = 0, gath = 0, nwords = 0;
	typedef typename RowVec<VEC>::type rowv;
	// a lane always folds the same word(s) of a row: (lane + 64 i) * VEC mod WD does not depend on i
	const rowv act = *reinterpret_cast<const rowv *>(active + ((lane * VEC) & (WD - 1)));
	for (int p = wave; p < n_parts; p += nwaves) {
		const int v0 = parts[2 * p], v1 = parts[2 * p + 1];
		const int nv = v1 - v0;
		const int e0 = (int)roff[v0], e1 = (int)roff[v1];
		u64 outdeg = 0; // only needed for new frontier vertices, but requested here, off the critical path
		if (lane < nv) outdeg = (u64)(off[v0 + lane + 1] - off[v0 + lane]);
		// Entry (lane, k) of a trip is in-slot base + 64*k + lane.  Two-stage software pipeline over two register sets
		// (A, B): while the records of one trip are in flight, the next trip's bit-map look-ups run and its records are
		// requested; a set's adjacency words are requested two trips ahead, right after the set's look-ups have read
		// them.  rpk is padded, entries past e1 are masked (their record fetch is out of range of the buffer
		// descriptor: no memory request, the same as cold entries).
		constexpr int T = 64 * UN;
		u32 ca[UN], cb[UN];
#pragma unroll
		for (int k = 0; k < UN; k++) ca[k] = rpk[e0 + 64 * k + lane];
#pragma unroll
		for (int k = 0; k < UN; k++) cb[k] = rpk[e0 + T + 64 * k + lane];
		for (int idx = lane; idx < nv * WD; idx += 64) acc[idx] = 0;
		if (lane < NV) nzn[lane] = 0;
		__builtin_amdgcn_wave_barrier();
		auto issue = [&](int base, u32 (&c)[UN], u32 (&kept)[UN], u32 (&q0)[UN], u32 (&q1)[UN], u32 (&q2)[UN], u32 (&q3)[UN],
		                 u32 &hotmask) {
			u64 bw[UN];
			u32 bb[UN];
#pragma unroll
			for (int k = 0; k < UN; k++) {
				kept[k] = c[k];
				const u32 b = (c[k] & 0x0FFFFFFFu) >> 6;
				if constexpr (LDSMAP) {
					bw[k] = s_bits[b];
					bb[k] = (u32)s_base[b] + s_super[b / kGroupBlocks];
				} else {
					const BlkInfo bi = blk[b];
					bw[k] = bi.bits;
					bb[k] = bi.base;
				}
			}
			hotmask = 0;
#pragma unroll
			for (int k = 0; k < UN; k++) {
				const u32 pos = kept[k] & 63u;
				const bool hot = ((bw[k] >> pos) & 1ull) && (base + 64 * k + lane < e1);
				hotmask |= hot ? 1u << k : 0u;
				const u32 idx = bb[k] + (u32)__popcll(bw[k] & ((1ull << pos) - 1ull));
				const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, hot ? idx * 16u : 0x80000000u, 0, 0);
				q0[k] = t[0];
				q1[k] = t[1];
				q2[k] = t[2];
				q3[k] = t[3];
			}
#pragma unroll
			for (int k = 0; k < UN; k++) c[k] = rpk[base + 2 * T + 64 * k + lane];
		};
		// Lane ids 0 and 1 of a record are set by the lane that fetched it.  Ids 2..9 exist for few entries (SF100
		// level 2: 16 % of the in-edges) but for at least one of nearly every 64, so a per-lane loop would run all ten
		// rounds at ~10 % lane use: those entries are queued in LDS instead (three tail words + owner row + count) and
		// served 64 at a time, one queue entry per lane (the queue fills across trips and is drained when full and at the
		// end of the part).
		u32 qn = 0; // queue fill, wave-uniform
		auto drain = [&]() {
			if (qn == 0) return;
			uint4 e = make_uint4(0, 0, 0, 0);
			if ((u32)lane < qn) e = queue[lane];
			TailLoop<2>::run(acc32, (e.w >> 8) * (2 * WD), e.x, e.y, e.z, e.w & 0xFFu);
			qn = 0;
		};
		auto consume = [&](const u32 (&kept)[UN], const u32 (&q0)[UN], const u32 (&q1)[UN], const u32 (&q2)[UN],
		                   const u32 (&q3)[UN], u32 hotmask) {
#pragma unroll
			for (int k = 0; k < UN; k++) {
				const bool hot = (hotmask >> k) & 1u;
				if (!__any(hot)) continue;
				const u32 own = kept[k] >> 28;
				gath += hot ? 1u : 0u;
				if constexpr (WD <= 2) {
					if (hot) {
						const u64 w0 = ((u64)q1[k] << 32) | q0[k];
						if (w0) atomicOr(&acc[own * WD], w0);
						if constexpr (WD == 2) {
							const u64 w1 = ((u64)q3[k] << 32) | q2[k];
							if (w1) atomicOr(&acc[own * WD + 1], w1);
						}
					}
				} else {
					const u32 cb8 = hot ? q3[k] >> 24 : 0u;
					const u32 count = cb8 == kOverflowCount ? 0u : cb8;
					const u32 ownbase = own * (2 * WD);
					if (count > 0) set_lane_bit(acc32, ownbase, q3[k] & 0xFFFu);
					if (count > 1) set_lane_bit(acc32, ownbase, (q3[k] >> 12) & 0xFFFu);
					const u64 tm = __ballot(count > 2);
					if (tm) {
						const u32 nt = (u32)__popcll(tm);
						if (qn + nt > (u32)QCAP) drain();
						if (count > 2) {
							const u32 pos = qn + __builtin_amdgcn_mbcnt_hi((u32)(tm >> 32), __builtin_amdgcn_mbcnt_lo((u32)tm, 0u));
							queue[pos] = make_uint4(q0[k], q1[k], q2[k], (own << 8) | count);
						}
						qn += nt;
					}
					// a neighbour carrying more lanes than a record holds: OR its dense row (WD contiguous words)
					u64 om = __ballot(cb8 == kOverflowCount);
					while (om) {
						const int j = __ffsll((long long)om) - 1;
						om &= om - 1;
						const u32 cj = (u32)__shfl((int)kept[k], j);
						if (lane < WD) {
							const u64 w = front[(size_t)(cj & 0x0FFFFFFFu) * WD + lane];
							if (w) atomicOr(&acc[(cj >> 28) * WD + lane], w);
						}
					}
				}
			}
		};
		u32 ka[UN], a0[UN], a1[UN], a2[UN], a3[UN], hota;
		u32 kb[UN], b0[UN], b1[UN], b2[UN], b3[UN], hotb;
		issue(e0, ca, ka, a0, a1, a2, a3, hota);
		for (int base = e0;; base += 2 * T) {
			// the look-ahead issue is unconditional: past e1 every lane is masked (no record request), whereas a
			// branch around it would make the compiler copy the in-flight registers at the join and wait for them
			issue(base + T, cb, kb, b0, b1, b2, b3, hotb);
			consume(ka, a0, a1, a2, a3, hota);
			if (base + T >= e1) break;
			issue(base + 2 * T, ca, ka, a0, a1, a2, a3, hota);
			consume(kb, b0, b1, b2, b3, hotb);
			if (base + 2 * T >= e1) break;
		}
		drain();
		scanned += (u64)(e1 - e0);
		__builtin_amdgcn_wave_barrier();
		// -- epilogue: fold into seen/next (contiguous rows, VEC words per lane), non-empty-word masks, frontier stats
		rowv sv[RPL], fresh[RPL], upd[RPL];
		u32 fbits[RPL];
#pragma unroll
		for (int i = 0; i < RPL; i++) { // the part's seen rows are contiguous: all requests go out before the first fold
			const int idx = (lane + 64 * i) * VEC;
			sv[i] = RowVec<VEC>::zero();
			if (idx < nv * WD) sv[i] = *reinterpret_cast<const rowv *>(seen + (size_t)v0 * WD + idx);
		}
#pragma unroll
		for (int i = 0; i < RPL; i++) { // all rows are folded before the first store: a store's source registers
			const int idx = (lane + 64 * i) * VEC; // cannot be reused until it has left, which would serialise the rows
			fbits[i] = 0;
			fresh[i] = RowVec<VEC>::zero();
			upd[i] = sv[i];
			if (idx < nv * WD) fbits[i] = RowVec<VEC>::fold(*reinterpret_cast<const rowv *>(acc + idx), act, sv[i], fresh[i], upd[i]);
		}
#pragma unroll
		for (int i = 0; i < RPL; i++) {
			const int idx = (lane + 64 * i) * VEC;
			if (idx < nv * WD) {
				if (fbits[i]) {
					*reinterpret_cast<rowv *>(seen + (size_t)v0 * WD + idx) = upd[i];
					atomicOr(&nzn[idx / WD], fbits[i] << (idx & (WD - 1)));
				}
				*reinterpret_cast<rowv *>(next + (size_t)v0 * WD + idx) = fresh[i];
			}
		}
		__builtin_amdgcn_wave_barrier();
		if (lane < nv) {
			const u32 fm = nzn[lane];
			nz_next[v0 + lane] = fm;
			if (fm) {
				nf += 1;
				nwords += (u64)__popc(fm);
				mf += outdeg;
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	for (int o = 32; o > 0; o >>= 1) {
		gath += __shfl_down(gath, o);
		nf += __shfl_down(nf, o);
		mf += __shfl_down(mf, o);
		nwords += __shfl_down(nwords, o);
	}
	if (lane == 0) {
		red[wib][0] = nf;
		red[wib][1] = mf;
		red[wib][2] = scanned;
		red[wib][3] = gath;
		red[wib][4] = nwords;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u64 a = 0, bsum = 0, c = 0, g = 0, wsum = 0;
#pragma unroll 1
		for (int k = 0; k < WPB; k++) {
			a += red[k][0];
			bsum += red[k][1];
			c += red[k][2];
			g += red[k][3];
			wsum += red[k][4];
		}
		if (wsum) atomicAdd(&cnt->front_words, (u32)wsum);
		if (a) atomicAdd(&cnt->front_vertices, (u32)a);
		if (bsum) atomicAdd(&cnt->front_edges, bsum);
		if (c) atomicAdd(&cnt->edges_scanned, c);
		if (g) atomicAdd(&cnt->word_gathers, g);
	}
}

// ---- host side -------------------------------------------------------------------------------------------------------

template <int WD, int UN>
static int launch_lanes(pgq_csr *c, Workspace *ws, const u64 *front, const u32 *nz, u64 *seen, u64 *next, u32 *nz_next,
                        const u64 *active, int stop, Counters *d_cnt) {
	const Options &opt = options();
	hipStream_t st = ws->stream;
	const int64_t V = c->V;
	const int ncu = device_cus();
	const int n_blk = (int)((V + 63) / 64);
	PGQ_TRY(ws->lblk.reserve((size_t)(n_blk + 1) * sizeof(BlkInfo)));
	PGQ_TRY(ws->lrec.reserve((size_t)std::max<int64_t>(V, 1) * sizeof(uint4)));
	u32 *d_total = reinterpret_cast<u32 *>(&d_cnt->pad); // records written (zeroed by k_level_reset)
	hipLaunchKernelGGL(k_compact_lanes<WD>, dim3(std::min(blocks_for(V / 8 + 1), 2u * ncu)), dim3(256), 0, st, nz, front, V,
	                   ws->lblk.as<BlkInfo>(), ws->lrec.as<uint4>(), d_total, stop, d_cnt);
	// the 1-bit frontier map and the record bases stay in LDS when they fit beside the accumulators
	auto kfn = k_pull_lanes<WD, UN, true>;
	static std::atomic<size_t> static_lds { 0 };
	size_t sl = static_lds.load();
	if (!sl) {
		hipFuncAttributes fa;
		sl = hipFuncGetAttributes(&fa, (const void *)kfn) == hipSuccess ? fa.sharedSizeBytes + 1 : 1;
		static_lds.store(sl);
	}
	const u32 rec_bytes = (u32)std::min<int64_t>(std::max<int64_t>(V, 1) * 16, 0x7FFFFFF0ll);
	const size_t dyn_bytes = (size_t)n_blk * 8 + (size_t)((n_blk + 1) & ~1) * 2 + (size_t)(n_blk / kGroupBlocks + 2) * 4 + 16;
	const bool lds_map = opt.sparse_lds && sl > 1 && sl + dyn_bytes + 256 <= 160 * 1024;
	if (lds_map) {
		static std::atomic<int> attr_set { 0 };
		if (!attr_set.load()) {
			(void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - sl));
			attr_set.store(1);
		}
		hipLaunchKernelGGL(kfn, dim3(ncu), dim3(1024), dyn_bytes, st, c->roff, c->rpk, c->off, c->pull_parts, c->n_pull_parts,
		                   ws->lblk.as<BlkInfo>(), ws->lrec.as<uint4>(), front, seen, next, nz_next, active, n_blk, rec_bytes, stop, d_cnt);
	} else {
		const unsigned grid = (unsigned)std::max(1, opt.blocks_per_cu) * ncu;
		hipLaunchKernelGGL((k_pull_lanes<WD, UN, false>), dim3(grid), dim3(256), 0, st, c->roff, c->rpk, c->off, c->pull_parts,
		                   c->n_pull_parts, ws->lblk.as<BlkInfo>(), ws->lrec.as<uint4>(), front, seen, next, nz_next, active,
		                   n_blk, rec_bytes, stop, d_cnt);
	}
	return PGQ_OK;
}

int pull_lanes_level(pgq_csr *c, Workspace *ws, int wd, const u64 *front, const u32 *nz, u64 *seen, u64 *next,
                     u32 *nz_next, const u64 *active, int stop, Counters *d_cnt) {
	const int un = options().lanes_unroll >= 4 ? 4 : (options().lanes_unroll >= 2 ? 2 : 1);
#define PGQ_LANES(W)                                                                                                   \
	case W:                                                                                                            \
		if (un == 4) return launch_lanes<W, 4>(c, ws, front, nz, seen, next, nz_next, active, stop, d_cnt);            \
		if (un == 2) return launch_lanes<W, 2>(c, ws, front, nz, seen, next, nz_next, active, stop, d_cnt);            \
		return launch_lanes<W, 1>(c, ws, front, nz, seen, next, nz_next, active, stop, d_cnt);
	switch (wd) {
		PGQ_LANES(1)
		PGQ_LANES(2)
		PGQ_LANES(4)
		PGQ_LANES(8)
		PGQ_LANES(16)
		PGQ_LANES(32)
	}
#undef PGQ_LANES
	return fail(PGQ_ERR_INVALID_ARG, "unsupported lane-word count");
}

} // namespace pgq

// tools/chunk_mt.cpp — native worker threads calling the chunk entry point concurrently, as DuckDB's expression executor does
// (iterativelength.cpp:34: once per DataChunk per worker thread).  Python threads cannot measure this: re-taking the GIL after
// every 70-us call turns into a convoy.  The CSR comes from files written by tools/chunk_throughput.py (offsets, adjacency).
//   g++ -O2 -std=c++17 -Iinclude -o tools/chunk_mt tools/chunk_mt.cpp -Lduckpgq-extension_amd/csrc -lpgq_hip -Wl,-rpath,'$ORIGIN/../duckpgq-extension_amd/csrc' -pthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "pgq_hip.h"

static std::vector<int64_t> read_i64(const char *path) {
	FILE *f = fopen(path, "rb");
	if (!f) { perror(path); exit(1); }
	fseek(f, 0, SEEK_END);
	const long n = ftell(f) / 8;
	fseek(f, 0, SEEK_SET);
	std::vector<int64_t> v((size_t)n);
	if (fread(v.data(), 8, (size_t)n, f) != (size_t)n) { perror("fread"); exit(1); }
	fclose(f);
	return v;
}

int main(int argc, char **argv) {
	if (argc < 3) { fprintf(stderr, "usage: chunk_mt offsets.bin adj.bin [chunks per thread]\n"); return 2; }
	const std::vector<int64_t> off = read_i64(argv[1]), adj = read_i64(argv[2]);
	const int64_t V = (int64_t)off.size() - 1;
	const int per_thread = argc > 3 ? atoi(argv[3]) : 300;
	pgq_csr_t *csr = nullptr;
	if (pgq_init(-1) != PGQ_OK || pgq_csr_upload(V, off.data(), adj.data(), nullptr, nullptr, PGQ_W_NONE, &csr) != PGQ_OK) {
		fprintf(stderr, "upload failed: %s\n", pgq_last_error());
		return 1;
	}
	const int64_t n = 2048;
	printf("{");
	bool first = true;
	for (int shape = 0; shape < 2; shape++)
		for (int T : { 1, 2, 4, 8, 16, 32, 64 }) {
			std::atomic<int> ready { 0 }, go { 0 }, bad { 0 };
			std::vector<double> secs((size_t)T, 0.0);
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++)
				th.emplace_back([&, t] {
					std::mt19937_64 rng(1000 + t);
					std::vector<int64_t> src((size_t)n), dst((size_t)n), out((size_t)n);
					std::vector<uint64_t> valid((size_t)n / 64 + 1);
					const int64_t one = (int64_t)(rng() % (uint64_t)V);
					for (int64_t i = 0; i < n; i++) {
						src[(size_t)i] = shape == 0 ? one : (int64_t)(rng() % (uint64_t)V);
						dst[(size_t)i] = (int64_t)(rng() % (uint64_t)V);
					}
					const pgq_vec_t sv { src.data(), nullptr, nullptr }, dv { dst.data(), nullptr, nullptr };
					for (int k = 0; k < 5; k++)
						if (pgq_iterativelength(csr, V, n, sv, dv, out.data(), valid.data()) != PGQ_OK) bad = 1;
					ready++;
					while (!go.load()) std::this_thread::yield();
					const auto t0 = std::chrono::steady_clock::now();
					for (int k = 0; k < per_thread; k++)
						if (pgq_iterativelength(csr, V, n, sv, dv, out.data(), valid.data()) != PGQ_OK) bad = 1;
					secs[(size_t)t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
					pgq_thread_release();
				});
			while (ready.load() < T) std::this_thread::yield();
			go = 1;
			for (auto &x : th) x.join();
			double wall = 0;
			for (double s : secs) wall = s > wall ? s : wall;
			if (bad) { fprintf(stderr, "a call failed: %s\n", pgq_last_error()); return 1; }
			printf("%s\"%s_T%d\": {\"rows_per_s\": %.0f, \"ms_per_chunk\": %.4f}", first ? "" : ", ", shape == 0 ? "cross_1src_x_2048dst" : "scattered_2048_pairs", T,
			       (double)T * per_thread * (double)n / wall, wall / per_thread * 1e3);
			first = false;
		}
	printf("}\n");
	pgq_csr_free(csr);
	return 0;
}

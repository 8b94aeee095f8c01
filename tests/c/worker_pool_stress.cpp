// worker_pool_stress.cpp -- TEST: the ingest path's WorkerPool (sdf-viewer_amd/host/worker_pool.hpp) under the pattern
// SDFViewer::update drives it with: many short sessions, a varying number of workers (threads are created mid-life), runs
// of a few microseconds back to back.  Every worker id of every run must execute exactly once; a lost wake-up hangs (the
// pytest wrapper runs this under a timeout).
#include <atomic>
#include <cstdio>
#include <vector>

#include "worker_pool.hpp"

int main(int argc, char** argv) {
    const unsigned max_workers = argc > 1 ? (unsigned)atoi(argv[1]) : 6;
    sdfviewer::WorkerPool pool;
    std::vector<std::atomic<unsigned>> hits(max_workers);
    unsigned long long runs = 0, state = 12345;
    for (int session = 0; session < 400; ++session) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned n = 1 + (unsigned)((state >> 33) % max_workers);  // grows and shrinks: bystanders + newcomers
        pool.begin(n);
        const int per_session = 1 + (int)((state >> 20) % 97);
        for (int r = 0; r < per_session; ++r) {
            const unsigned m = 1 + (unsigned)((state >> (r % 13)) % n);  // a run may use fewer workers than the session woke
            for (auto& h : hits) h.store(0);
            std::function<void(unsigned)> fn = [&](unsigned t) { hits[t].fetch_add(1); };
            pool.run(m, fn);
            for (unsigned t = 0; t < max_workers; ++t)
                if (hits[t].load() != (t < m ? 1u : 0u)) {
                    fprintf(stderr, "session %d run %d: worker %u ran %u times (m = %u)\n", session, r, t, hits[t].load(), m);
                    return 1;
                }
            ++runs;
        }
        pool.end();
    }
    printf("ok %llu runs\n", runs);
    return 0;
}

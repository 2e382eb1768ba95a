// Exercises bmm-mcmc_amd/csrc/host_crew.h the way a *_run call does -- one crew, an asynchronous job (the pack)
// while the caller goes on, then many short blocking jobs (the trace blocks), crews created and destroyed in a
// row, a crew that is destroyed without ever getting a job, a job still running when the crew goes away --
// and checks that every index of every job is visited exactly once.  Built by tests/test_host_crew.py, plainly and
// with -fsanitize=thread.
#include <cstdio>
#include <numeric>
#include <vector>

#include "host_crew.h"

using bmm_host::HostCrew;

static int fail(const char* what, long a, long b) {
    fprintf(stderr, "FAIL %s: %ld vs %ld\n", what, a, b);
    return 1;
}

int main() {
    if (bmm_host::host_threads() < 1 || bmm_host::host_threads() > 16) return fail("host_threads", bmm_host::host_threads(), 0);
    for (int round = 0; round < 20; ++round) {
        HostCrew crew;
        const int64_t n = 100003 + 977 * round;
        std::vector<int> hits((size_t)n, 0);
        std::vector<long> sums(64, 0);
        std::atomic<long> total{0};
        int* const h = hits.data();
        crew.begin(n, 1024, 64, [h, &total](int64_t lo, int64_t hi) {  // asynchronous, as AsyncPack starts it
            long s = 0;
            for (int64_t i = lo; i < hi; ++i) { h[i] += 1; s += i; }
            total.fetch_add(s);
        });
        long busy = 0;
        for (int i = 0; i < 1000; ++i) busy += i;  // the caller goes on
        crew.wait();
        for (int64_t i = 0; i < n; ++i) if (hits[(size_t)i] != 1) return fail("async job coverage", (long)i, hits[(size_t)i]);
        if (total.load() != (long)(n * (n - 1) / 2)) return fail("async job sum", total.load(), (long)(n * (n - 1) / 2));
        for (int blk = 0; blk < 50; ++blk) {  // blocking jobs in a row, as trace_out issues them
            const int64_t m = blk % 7 == 0 ? 5 : 4096 + 131 * blk;
            crew.run(m, 256, 64, [h](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) h[i] += 1; });
            for (int64_t i = 0; i < m; ++i) if (hits[(size_t)i] != blk + 2) return fail("blocking job coverage", (long)i, hits[(size_t)i]);
            for (int64_t i = m; i < m + 3 && i < n; ++i) if (hits[(size_t)i] > blk + 1) return fail("job ran past its end", (long)i, hits[(size_t)i]);
            for (int64_t i = 0; i < m; ++i) hits[(size_t)i] = blk + 2;
            for (int64_t i = m; i < n; ++i) hits[(size_t)i] = blk + 2;
        }
        crew.run(0, 256, 64, [](int64_t, int64_t) {});  // an empty job
        (void)busy;
    }
    for (int i = 0; i < 50; ++i) { HostCrew idle; }  // never given a job
    {
        std::vector<int> v(1 << 20, 0);
        int* const p = v.data();
        {
            HostCrew crew;
            crew.begin((int64_t)v.size(), 4096, 64, [p](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) p[i] = 1; });
        }  // the destructor waits for the job
        if (std::accumulate(v.begin(), v.end(), 0L) != (long)v.size()) return fail("job finished by the destructor", 0, 0);
    }
    puts("ok");
    return 0;
}

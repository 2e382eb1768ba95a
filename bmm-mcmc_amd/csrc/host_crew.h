// bmm-mcmc on MI355X: the host threads behind the two ends of a *_run call (plain C++, no HIP): how many CPUs
// this process may use, and HostCrew, the threads of one call.  In a header of its own so that
// tests/test_host_crew.py can build it with -fsanitize=thread.
#pragma once
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace bmm_host {

// CPUs this process may use for the host-side ends of a run (packing X on its way in, widening the label
// trace on its way out): the affinity mask, cut to a cgroup CPU quota where there is one (a container that
// is granted 16 of a host's 256 hardware threads), at most 16 -- memory bandwidth is what those loops need.
inline int host_threads() {
    static const int n = [] {
        int n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
        long long quota = -1, period = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32] = "";
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // v1
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &period) != 1) period = 0;
                fclose(h);
            }
        }
        if (quota > 0 && period > 0) {
            const int c = (int)((quota + period - 1) / period);
            if (c >= 1 && c < n) n = c;
        }
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }();
    return n;
}

// The host threads of one call, started once and handed one range job after another (a *_run call packs X
// with them on its way in and widens the label trace, block after block, on its way out: starting sixteen
// threads one after the other costs 0.3 - 0.4 ms, and a trace of 270 blocks would pay that 270 times).
// A job is f(lo, hi) over pieces of [0, n) that the threads claim one by one; begin() returns at once, wait()
// takes pieces itself until none is left and returns when all are done.  f must not throw.  The constructor
// starts one thread, which starts two more and so on, so that the caller goes on after 30 us; a thread that
// cannot be started just leaves the others (in the end the caller in wait()) more pieces.  The threads end
// with the object: nothing outlives the call (a host process that forks, as R's parallel package does, finds
// no parked threads of ours).
class HostCrew {
    struct Job {
        std::function<void(int64_t, int64_t)> f;
        int64_t n = 0, per = 0, parts = 0;
        std::atomic<int64_t> next{0};
        int64_t done = 0;  // under m
    };
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> th;
    std::shared_ptr<Job> cur;
    uint64_t gen = 0;
    bool quit = false;
    void spawn(size_t i) noexcept {
        if (i >= th.size()) return;
        try {
            th[i] = std::thread([this, i]() { work(i); });
        } catch (...) {
        }
    }
    void take_pieces(Job& j) {
        for (;;) {
            const int64_t k = j.next.fetch_add(1, std::memory_order_relaxed);
            if (k >= j.parts) return;
            const int64_t lo = k * j.per, hi = lo + j.per < j.n ? lo + j.per : j.n;
            j.f(lo, hi);
            std::lock_guard<std::mutex> lk(m);
            if (++j.done == j.parts) cv_done.notify_all();
        }
    }
    void work(size_t i) {
        spawn(2 * i + 1);
        spawn(2 * i + 2);
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return quit || gen != seen; });
                if (gen == seen) return;  // quit, nothing pending
                seen = gen;
                j = cur;
            }
            if (j) take_pieces(*j);
        }
    }
public:
    HostCrew() {
        try {
            th.resize((size_t)host_threads());
        } catch (...) {
        }
        spawn(0);
    }
    ~HostCrew() {
        wait();
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_job.notify_all();
        // in index order: a thread has started its two successors before it looks for work, so after joining
        // thread i the slots 2i+1 and 2i+2 are final
        for (std::thread& t : th) if (t.joinable()) t.join();
    }
    HostCrew(const HostCrew&) = delete;
    HostCrew& operator=(const HostCrew&) = delete;
    template <class F>
    void begin(int64_t count, int64_t min_per_piece, int64_t align, F f) {
        wait();
        if (count <= 0) return;
        auto j = std::make_shared<Job>();
        int64_t parts = min_per_piece > 0 ? count / min_per_piece : 1;
        const int64_t most = 4 * (int64_t)(th.empty() ? 1 : th.size());
        parts = parts < 1 ? 1 : (parts > most ? most : parts);
        int64_t per = (count + parts - 1) / parts;
        per = (per + align - 1) / align * align;
        j->f = std::move(f);
        j->n = count;
        j->per = per;
        j->parts = (count + per - 1) / per;
        std::lock_guard<std::mutex> lk(m);
        cur = std::move(j);
        ++gen;
        cv_job.notify_all();
    }
    void wait() {
        std::shared_ptr<Job> j;
        { std::lock_guard<std::mutex> lk(m); j = cur; }
        if (!j) return;
        take_pieces(*j);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return j->done == j->parts; });
        if (cur == j) cur.reset();
    }
    template <class F>
    void run(int64_t count, int64_t min_per_piece, int64_t align, F f) { begin(count, min_per_piece, align, std::move(f)); wait(); }
};

}  // namespace bmm_host

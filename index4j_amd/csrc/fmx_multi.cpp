// fmx_multi.cpp — one immutable index on several GPUs behind the C ABI (include/fmx.h "replicas"): the sharded forms of the
// batch entry points.  FmIndex is @ThreadSafe and immutable (FM:82; one index per thread in the reference's own throughput
// benchmark, FmIndexThroughputState.java:30), every query of a batch is an independent read: a batch is cut into contiguous
// shards (fmx_shard_range), shard r runs the SINGLE-index entry point on replica r from a host thread of its own, and stores
// into its own slice of the caller's arrays.  No collective, no exchange on the query path (SURVEY 8e scheme (i)); the image
// itself travels once, in fmx_replicate (fmx_api.cpp).  Everything here sits on top of the public C ABI.
#include "../../include/fmx.h"

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace fmx {
int api_fail(int code, const std::string &msg);  // fmx_api.cpp: sets the calling thread's fmx_last_error
}

namespace {

// A call's shards complete on this: counted down by the workers, waited for by the calling thread (whose frame it lives in).
struct Latch {
    std::mutex m;
    std::condition_variable cv;
    int open;
    explicit Latch(int n) : open(n) {}
    void done() {
        std::lock_guard<std::mutex> lock(m);
        if (--open == 0) cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return open == 0; });
    }
};

struct Job {
    std::function<int()> run;
    int rc = FMX_OK;
    std::string err;
    Latch *latch = nullptr;
};

// One worker thread per (device, slot): slot = how many replicas of the call before this one live on the same device, so two
// replicas sharing a GPU (tests: devices {0, 0}) still run side by side.  A worker binds to its device once and keeps the
// stream sets the host-buffer entry points make per (thread, device).  Workers are started on first use and never stopped: like
// the library's other per-thread state they must outlive every thread_local destructor and the HIP runtime's own shutdown order
// (a JVM unloads nothing), so the table is leaked on purpose and the threads are detached.
struct Worker {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job *> queue;
    int device;
    explicit Worker(int d) : device(d) {}
    void loop() {
        (void)hipSetDevice(device);
        for (;;) {
            Job *job;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return !queue.empty(); });
                job = queue.front();
                queue.pop_front();
            }
            int rc;
            try {
                rc = job->run();
            } catch (...) {  // (the entry points do not throw; a std::function allocation might)
                rc = fmx::api_fail(FMX_E_UNSUPPORTED, "internal error in a shard's worker");
            }
            job->rc = rc;
            if (rc != FMX_OK) job->err = fmx_last_error();
            Latch *latch = job->latch;
            latch->done();  // (the job belongs to the caller's frame: not touched after this)
        }
    }
    void submit(Job *job) {
        {
            std::lock_guard<std::mutex> lock(m);
            queue.push_back(job);
        }
        cv.notify_one();
    }
};

struct Workers {
    std::mutex m;
    std::map<std::pair<int, int>, Worker *> table;
};
Workers &workers() {
    static Workers *w = new Workers();  // leaked on purpose (see above)
    return *w;
}
Worker *worker_for(int device, int slot) {
    Workers &w = workers();
    std::lock_guard<std::mutex> lock(w.m);
    Worker *&p = w.table[{device, slot}];
    if (!p) {
        p = new Worker(device);
        std::thread(&Worker::loop, p).detach();
    }
    return p;
}

int check_replicas(const fmx_index *const *replicas, int32_t n_replicas, int32_t stride = 1) {
    if (!replicas || n_replicas < 1 || stride < 1) return fmx::api_fail(FMX_E_ARG, "no replicas");
    for (int32_t r = 0; r < n_replicas; ++r) {
        const fmx_index *idx = replicas[(size_t)r * (size_t)stride];
        if (!idx) return fmx::api_fail(FMX_E_ARG, "null replica");
        if (fmx_device_of(idx) < 0) return fmx::api_fail(FMX_E_NO_DEVICE, "replica is not resident on a HIP device");
    }
    return FMX_OK;
}

// body(r) for every replica r in `active` at once — r = active[0] on the calling thread, the others on their devices' workers —
// and the first failure in replica order (its message becomes this thread's fmx_last_error).
int run_on_replicas(const fmx_index *const *replicas, int32_t stride, const std::vector<int32_t> &active,
                    const std::function<int(int32_t)> &body) {
    if (active.empty()) return FMX_OK;
    std::vector<Job> jobs(active.size());
    Latch latch((int)active.size() - 1);
    std::map<int, int> slots;  // device -> replicas of this call seen on it so far
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);
    for (size_t k = 0; k < active.size(); ++k) {
        const int32_t r = active[k];
        const int device = fmx_device_of(replicas[(size_t)r * (size_t)stride]);
        const int slot = slots[device]++;
        jobs[k].run = [&body, r]() { return body(r); };
        jobs[k].latch = &latch;
        if (k > 0) worker_for(device, slot)->submit(&jobs[k]);
    }
    {
        const int device = fmx_device_of(replicas[(size_t)active[0] * (size_t)stride]);
        (void)hipSetDevice(device);
        jobs[0].rc = jobs[0].run();
        if (jobs[0].rc != FMX_OK) jobs[0].err = fmx_last_error();
        (void)hipSetDevice(caller_device);
    }
    latch.wait();
    for (size_t k = 0; k < jobs.size(); ++k)
        if (jobs[k].rc != FMX_OK)
            return fmx::api_fail(jobs[k].rc, "shard " + std::to_string(active[k]) + " (device " +
                                                 std::to_string(fmx_device_of(replicas[(size_t)active[k] * (size_t)stride])) +
                                                 "): " + jobs[k].err);
    return FMX_OK;
}

// shards of a host batch: body(r, lo, hi) for every replica whose shard is not empty
int run_sharded(const fmx_index *const *replicas, int32_t n_replicas, int32_t stride, int32_t n,
                const std::function<int(int32_t, int32_t, int32_t)> &body) {
    int rc = check_replicas(replicas, n_replicas, stride);
    if (rc) return rc;
    if (n < 0) return fmx::api_fail(FMX_E_ARG, "bad arguments");
    std::vector<int32_t> active;
    for (int32_t r = 0; r < n_replicas; ++r) {
        int64_t lo, hi;
        fmx_shard_range(n, n_replicas, r, &lo, &hi);
        if (hi > lo) active.push_back(r);
    }
    return run_on_replicas(replicas, stride, active, [&](int32_t r) {
        int64_t lo, hi;
        fmx_shard_range(n, n_replicas, r, &lo, &hi);
        return body(r, (int32_t)lo, (int32_t)hi);
    });
}

template <class F>
int guarded_multi(F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fmx::api_fail(FMX_E_UNSUPPORTED, "out of host memory");
    } catch (const std::exception &e) {
        return fmx::api_fail(FMX_E_UNSUPPORTED, std::string("internal error: ") + e.what());
    }
}

template <class T>
T *at(T *p, int64_t i) {
    return p ? p + i : nullptr;
}

}  // namespace

extern "C" {

void fmx_shard_range(int64_t n, int32_t parts, int32_t part, int64_t *lo, int64_t *hi) {
    int64_t a = 0, b = 0;
    if (n > 0 && parts > 0 && part >= 0 && part < parts) {
        const int64_t base = n / parts, rem = n % parts;
        a = (int64_t)part * base + (part < rem ? part : rem);
        b = a + base + (part < rem ? 1 : 0);
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
}

int fmx_count_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const uint16_t *pat, const int32_t *pat_off,
                          int32_t n, int32_t *counts, int32_t *lf_steps, int32_t *status) {
    return guarded_multi([&]() -> int {
        if (n > 0 && (!pat_off || !counts)) return fmx::api_fail(FMX_E_ARG, "bad arguments");
        return run_sharded(replicas, n_replicas, 1, n, [&](int32_t r, int32_t lo, int32_t hi) {
            return fmx_count_batch(replicas[r], pat, pat_off + lo, hi - lo, counts + lo, at(lf_steps, lo), at(status, lo));
        });
    });
}

int fmx_locate_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const uint16_t *pat, const int32_t *pat_off,
                           int32_t n, int32_t max_matches, int32_t *locs, int32_t loc_cap, int32_t *found, int32_t *lf_steps,
                           int32_t *status) {
    return guarded_multi([&]() -> int {
        if (loc_cap < 0 || (n > 0 && (!pat_off || !found || (!locs && loc_cap > 0)))) return fmx::api_fail(FMX_E_ARG, "bad arguments");
        return run_sharded(replicas, n_replicas, 1, n, [&](int32_t r, int32_t lo, int32_t hi) {
            return fmx_locate_batch(replicas[r], pat, pat_off + lo, hi - lo, max_matches, at(locs, (int64_t)lo * loc_cap), loc_cap,
                                    found + lo, at(lf_steps, lo), at(status, lo));
        });
    });
}

int fmx_extract_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const int32_t *start, const int32_t *stop,
                            int32_t n, uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf_steps,
                            int32_t *status) {
    return guarded_multi([&]() -> int {
        if (dst_len < 0 || (n > 0 && (!start || !stop || !out_len || (!dst && dst_len > 0)))) return fmx::api_fail(FMX_E_ARG, "bad arguments");
        return run_sharded(replicas, n_replicas, 1, n, [&](int32_t r, int32_t lo, int32_t hi) {
            return fmx_extract_batch(replicas[r], start + lo, stop + lo, hi - lo, at(dst, (int64_t)lo * dst_len), dst_len, offset,
                                     out_len + lo, at(lf_steps, lo), at(status, lo));
        });
    });
}

int fmx_extract_boundary_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const int32_t *from, int32_t n,
                                     uint16_t boundary, int mode, uint16_t *dst, int32_t dst_len, int32_t offset,
                                     int32_t *out_len, int32_t *lf_steps, int32_t *status, int32_t *aux) {
    return guarded_multi([&]() -> int {
        if (dst_len < 0 || mode < 0 || mode > 2 || (n > 0 && (!from || !out_len || (!dst && dst_len > 0))))
            return fmx::api_fail(FMX_E_ARG, "bad arguments");
        return run_sharded(replicas, n_replicas, 1, n, [&](int32_t r, int32_t lo, int32_t hi) {
            return fmx_extract_boundary_batch(replicas[r], from + lo, hi - lo, boundary, mode, at(dst, (int64_t)lo * dst_len), dst_len,
                                              offset, out_len + lo, at(lf_steps, lo), at(status, lo), at(aux, lo));
        });
    });
}

int fmx_count_locate_segments_multi(const fmx_index *const *segs, int32_t n_replicas, int32_t n_segs, const int64_t *seg_base,
                                    const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t max_matches, int64_t *counts,
                                    int64_t *lf_steps, int64_t *locs, int32_t *found, int32_t *status) {
    return guarded_multi([&]() -> int {
        if (n_segs < 1 || max_matches < 1 || !seg_base || (n > 0 && (!pat_off || !counts || !locs || !found)))
            return fmx::api_fail(FMX_E_ARG, "bad arguments");
        return run_sharded(segs, n_replicas, n_segs, n, [&](int32_t r, int32_t lo, int32_t hi) {
            return fmx_count_locate_segments(segs + (size_t)r * (size_t)n_segs, n_segs, seg_base, pat, pat_off + lo, hi - lo, max_matches,
                                             counts + lo, at(lf_steps, lo), locs + (int64_t)lo * max_matches, found + lo, at(status, lo));
        });
    });
}

// ---- device-resident shards: every replica's launches issued at once, nothing waited for ----

int fmx_count_batch_multi_dev(const fmx_index *const *replicas, int32_t n_replicas, const uint16_t *const *d_pat,
                              const int32_t *const *d_pat_off, const int32_t *n, int32_t *const *d_counts,
                              int32_t *const *d_lf_steps, int32_t *const *d_status, void *const *streams) {
    return guarded_multi([&]() -> int {
        int rc = check_replicas(replicas, n_replicas);
        if (rc) return rc;
        if (!d_pat || !d_pat_off || !n || !d_counts) return fmx::api_fail(FMX_E_ARG, "bad arguments");
        std::vector<int32_t> active;
        for (int32_t r = 0; r < n_replicas; ++r)
            if (n[r] > 0) active.push_back(r);
        return run_on_replicas(replicas, 1, active, [&](int32_t r) {
            return fmx_count_batch_dev(replicas[r], d_pat[r], d_pat_off[r], n[r], d_counts[r], d_lf_steps ? d_lf_steps[r] : nullptr,
                                       d_status ? d_status[r] : nullptr, streams ? streams[r] : nullptr);
        });
    });
}

int fmx_count_locate_segments_multi_dev(const fmx_index *const *segs, int32_t n_replicas, int32_t n_segs, const int64_t *seg_base,
                                        const uint16_t *const *d_pat, const int32_t *const *d_pat_off, const int32_t *n,
                                        int32_t max_matches, int64_t *const *d_counts, int64_t *const *d_lf_steps,
                                        int64_t *const *d_locs, int32_t *const *d_found, int32_t *const *d_status,
                                        int32_t *const *d_tmp, void *const *streams) {
    return guarded_multi([&]() -> int {
        int rc = check_replicas(segs, n_replicas, n_segs);
        if (rc) return rc;
        if (!d_pat || !d_pat_off || !n || !d_counts || !d_locs || !d_found || !d_tmp || !seg_base)
            return fmx::api_fail(FMX_E_ARG, "bad arguments");
        std::vector<int32_t> active;
        for (int32_t r = 0; r < n_replicas; ++r)
            if (n[r] > 0) active.push_back(r);
        return run_on_replicas(segs, n_segs, active, [&](int32_t r) {
            return fmx_count_locate_segments_dev(segs + (size_t)r * (size_t)n_segs, n_segs, seg_base, d_pat[r], d_pat_off[r], n[r],
                                                 max_matches, d_counts[r], d_lf_steps ? d_lf_steps[r] : nullptr, d_locs[r], d_found[r],
                                                 d_status ? d_status[r] : nullptr, d_tmp[r], streams ? streams[r] : nullptr);
        });
    });
}

int fmx_multi_synchronize(const fmx_index *const *replicas, int32_t n_replicas, void *const *streams) {
    return guarded_multi([&]() -> int {
        int rc = check_replicas(replicas, n_replicas);
        if (rc) return rc;
        int caller_device = 0;
        (void)hipGetDevice(&caller_device);
        int first = FMX_OK;
        std::string err;
        for (int32_t r = 0; r < n_replicas; ++r) {
            hipError_t e = hipSetDevice(fmx_device_of(replicas[r]));
            if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(streams ? streams[r] : nullptr));
            if (e != hipSuccess && first == FMX_OK) {
                (void)hipGetLastError();
                first = FMX_E_HIP;
                err = "replica " + std::to_string(r) + ": " + hipGetErrorString(e);
            }
        }
        (void)hipSetDevice(caller_device);
        return first == FMX_OK ? FMX_OK : fmx::api_fail(first, err);
    });
}

}  // extern "C"

// xm_pool.h -- worker pool shared by the host-side stripper / writer (xm_sam.cpp) and the BAM decoder (xm_bam.cpp).
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace xmh {

// Workers that live as long as the parser: a window needs about fifteen short parallel phases, and starting sixteen
// threads for each of them costs more than several of the phases themselves.
class Pool {
public:
    explicit Pool(int n) : n_(std::max(1, n))
    {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t]() { loop(t); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        wake_.notify_all();
        for (auto &th : workers_) th.join();
    }
    Pool(const Pool &) = delete;
    Pool &operator=(const Pool &) = delete;
    int size() const { return n_; }

    // fn(t) for t in [0, n_tasks), n_tasks <= size(); task 0 runs on the calling thread.  The first exception thrown
    // by a task is rethrown here once every task has finished.
    void run(int n_tasks, const std::function<void(int)> &fn)
    {
        n_tasks = std::min(n_tasks, n_);
        if (n_tasks <= 0) return;
        if (n_tasks == 1) {
            fn(0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            tasks_ = n_tasks;
            pending_ = n_tasks - 1;
            error_ = nullptr;
            ++gen_;
        }
        wake_.notify_all();
        std::exception_ptr mine;
        try { fn(0); } catch (...) { mine = std::current_exception(); }
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        job_ = nullptr;
        tasks_ = 0;
        std::exception_ptr err = mine ? mine : error_;
        error_ = nullptr;
        lk.unlock();
        if (err) std::rethrow_exception(err);
    }

private:
    void loop(int t)
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            wake_.wait(lk, [&]() { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            if (t >= tasks_ || !job_) continue;
            const std::function<void(int)> *job = job_;
            lk.unlock();
            std::exception_ptr err;
            try { (*job)(t); } catch (...) { err = std::current_exception(); }
            lk.lock();
            if (err && !error_) error_ = err;
            if (--pending_ == 0) done_.notify_one();
        }
    }

    const int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(int)> *job_ = nullptr;
    int tasks_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::exception_ptr error_;
};

// fn(worker, begin, end) over [0, n) in one contiguous slice per worker
template <typename F>
void parallel_for(Pool &pool, uint64_t n, F fn, int max_workers = 1 << 30)
{
    const int workers = std::min(pool.size(), max_workers);
    if (workers <= 1 || n < 4096) {
        fn(0, (uint64_t)0, n);
        return;
    }
    const uint64_t per = (n + (uint64_t)workers - 1) / (uint64_t)workers;
    const int used = (int)((n + per - 1) / per);
    pool.run(used, [&](int t) {
        const uint64_t b = std::min<uint64_t>(n, (uint64_t)t * per), e = std::min<uint64_t>(n, b + per);
        if (b < e) fn(t, b, e);
    });
}

}  // namespace xmh

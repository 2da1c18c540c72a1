// throttlecrab_actor.hpp -- the batch-draining actor (C++17, header only) over the host
// mirror in throttlecrab_gpu.hpp: SURVEY.md section 8(f) row 1.
//
// Mirrors, with the reference's names and behaviour:
//   ThrottleRequest / ThrottleResponse        throttlecrab-server/src/types.rs:32-96
//       (reset_after / retry_after are whole seconds, truncated: types.rs:87-96)
//   RateLimiterMessage::Throttle               throttlecrab-server/src/actor.rs:35-45
//   RateLimiterHandle::throttle                throttlecrab-server/src/actor.rs:52-82
//   RateLimiterActor::spawn_*                  throttlecrab-server/src/actor.rs:88-168
//   run_actor / handle_throttle                throttlecrab-server/src/actor.rs:217-255
//
// What changes against the reference: its actor takes ONE message per loop turn
// (`while let Some(msg) = rx.recv().await`, actor.rs:222) and calls rate_limit once; this one
// takes EVERYTHING that is queued (tokio's `recv_many`), in queue order, and hands it to
// RateLimiter::submit_batch / collect_batch (the pipelined form of rate_limit_batch: up to three
// batches in flight while the next one is drained and marshalled), whose result is by
// construction the result of the one-by-one loop.  Requests keep the timestamp their transport stamped (types.rs:46), so a
// batch carries per-request, possibly non-monotone `now` values, like the reference's queue.
//
// Channel semantics follow tokio's bounded mpsc: `throttle` blocks while the buffer is full
// (back-pressure, actor.rs:71-77), the actor exits when every handle is gone (actor.rs:222,235),
// a send after shutdown fails with "Rate limiter actor has shut down" (actor.rs:77).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <variant>
#include <vector>

#include "throttlecrab_gpu.hpp"

namespace throttlecrab {
namespace server {

// anyhow::Result<T>: the value or the error's Display string
template <class T>
using Result = std::variant<T, std::string>;
template <class T>
inline bool is_ok(const Result<T>& r) { return r.index() == 0; }

// types.rs:32-47
struct ThrottleRequest {
    std::string key;
    int64_t max_burst;
    int64_t count_per_period;
    int64_t period;
    int64_t quantity;
    SystemTime timestamp;
};

// types.rs:71-83
struct ThrottleResponse {
    bool allowed;
    int64_t limit;
    int64_t remaining;
    int64_t reset_after; // seconds
    int64_t retry_after; // seconds
    // impl From<(bool, RateLimitResult)> (types.rs:85-96): Duration::as_secs() truncates
    static ThrottleResponse from(bool allowed, const RateLimitResult& r) {
        return ThrottleResponse{allowed, r.limit, r.remaining, r.reset_after.count() / 1000000000LL,
                                r.retry_after.count() / 1000000000LL};
    }
};

// actor.rs:35-45.  A message carries one request and its oneshot (the reference's Throttle variant) or --
// `many` non-empty -- a connection's whole pipelined buffer with one oneshot for all replies, so that a
// transport pays for the channel once per buffer instead of once per request.
struct RateLimiterMessage {
    ThrottleRequest request;
    std::promise<Result<ThrottleResponse>> response_tx;
    // the group form lives behind one pointer: a std::promise allocates its shared state when it is constructed,
    // and single requests should not pay for a second one
    struct Many {
        std::vector<ThrottleRequest> requests;
        std::promise<std::vector<Result<ThrottleResponse>>> tx;
    };
    std::unique_ptr<Many> many;
    size_t size() const { return many ? many->requests.size() : 1; }
};

namespace detail {
// bounded multi-producer single-consumer channel + the actor thread that drains it.
// Round 5: the queue is SHARDED.  One deque under one mutex carried ~1 M messages/s however many threads sent (16 producers
// handing the lock to each other through the kernel -- tools/actor_bench.cpp, single requests: the reference's own message
// shape); now every handle sends into the shard it was given when it was made (round robin), under that shard's lock, and the
// actor drains all shards in turn.  What a channel guarantees is kept: the messages of ONE handle are evaluated in the order
// they were sent (one handle, one shard, FIFO); messages of different handles were never ordered against each other by
// anything but the race for the lock.  The bound on queued requests, senders blocking while it is reached, and the actor
// sleeping on an empty channel go through one counter and one mutex that is only taken to sleep or to wake somebody.
constexpr size_t SHARDS = 16;
struct alignas(64) Shard {
    std::mutex mu;
    std::deque<RateLimiterMessage> queue;
};
struct Channel {
    Shard shard[SHARDS];
    std::atomic<size_t> queued_requests{0}; // requests in the shards or about to be pushed (a message may carry many)
    std::atomic<uint32_t> next_shard{0};
    std::atomic<bool> closed{false}, actor_sleeping{false};
    std::atomic<uint32_t> senders_waiting{0};
    size_t buffer_size = 1;
    std::mutex mu; // sleeping and waking, the sender count, the statistics
    std::condition_variable not_empty, not_full;
    size_t senders = 0;
    std::thread actor;

    // statistics of the drain loop (how well the queue batches)
    uint64_t batches = 0, requests = 0, largest_batch = 0;
    // where the actor thread's time goes (ns; written by the actor thread only, read through RateLimiterHandle::loop_ns):
    // handing batches to the limiter, answering (of which: collect_batch, i.e. waiting for the GPU + the outcomes), releasing
    // answered messages
    std::atomic<uint64_t> ns_submit{0}, ns_answer{0}, ns_collect{0}, ns_release{0};
};
} // namespace detail

// actor.rs:52-82.  Copyable; the actor shuts down when the last copy is destroyed.
class RateLimiterHandle {
  public:
    using request_type = ThrottleRequest;
    RateLimiterHandle() = default;
    RateLimiterHandle(const RateLimiterHandle& o) : ch_(o.ch_) { retain(); } // (a copy is another sender: it gets a shard of its own)
    RateLimiterHandle(RateLimiterHandle&& o) noexcept : ch_(std::move(o.ch_)), shard_(o.shard_) {}
    RateLimiterHandle& operator=(RateLimiterHandle o) {
        std::swap(ch_, o.ch_);
        std::swap(shard_, o.shard_);
        return *this;
    }
    ~RateLimiterHandle() { release(); }

    // send + await (actor.rs:68-82).  Blocks while the buffer is full.
    Result<ThrottleResponse> throttle(ThrottleRequest request) { return throttle_async(std::move(request)).get(); }

    // the two halves separately: a transport thread can queue many requests before it waits
    std::future<Result<ThrottleResponse>> throttle_async(ThrottleRequest request) {
        RateLimiterMessage msg;
        msg.request = std::move(request);
        std::future<Result<ThrottleResponse>> rx = msg.response_tx.get_future();
        if (!send(msg)) msg.response_tx.set_value(std::string("Rate limiter actor has shut down"));
        return rx;
    }

    // A whole buffer of requests (a connection's pipelined commands) as ONE message: evaluated in order,
    // answered together.  Empty input -> empty output.
    std::future<std::vector<Result<ThrottleResponse>>> throttle_many_async(std::vector<ThrottleRequest> requests) {
        RateLimiterMessage msg;
        msg.many = std::make_unique<RateLimiterMessage::Many>();
        std::future<std::vector<Result<ThrottleResponse>>> rx = msg.many->tx.get_future();
        const size_t n = requests.size();
        if (n == 0) {
            msg.many->tx.set_value({});
            return rx;
        }
        msg.many->requests = std::move(requests);
        if (!send(msg)) msg.many->tx.set_value(std::vector<Result<ThrottleResponse>>(n, std::string("Rate limiter actor has shut down")));
        return rx;
    }
    std::vector<Result<ThrottleResponse>> throttle_many(std::vector<ThrottleRequest> requests) {
        return throttle_many_async(std::move(requests)).get();
    }

    // (batches drained, requests served, largest batch) so far
    std::tuple<uint64_t, uint64_t, uint64_t> drain_stats() const {
        std::lock_guard<std::mutex> lk(ch_->mu);
        return {ch_->batches, ch_->requests, ch_->largest_batch};
    }

    // (submit, answer, of which collect_batch, release) nanoseconds of the actor thread so far -- a diagnostic
    std::tuple<uint64_t, uint64_t, uint64_t, uint64_t> loop_ns() const {
        return {ch_->ns_submit.load(std::memory_order_relaxed), ch_->ns_answer.load(std::memory_order_relaxed), ch_->ns_collect.load(std::memory_order_relaxed),
                ch_->ns_release.load(std::memory_order_relaxed)};
    }

  private:
    template <class L>
    friend class BasicRateLimiterActor;
    explicit RateLimiterHandle(std::shared_ptr<detail::Channel> ch) : ch_(std::move(ch)) { retain(); }
    // false: the actor has shut down (msg is untouched).  Blocks while the buffer is full (a message
    // larger than the whole buffer is let in when the buffer is empty).
    bool send(RateLimiterMessage& msg) {
        if (!ch_) return false;
        detail::Channel& ch = *ch_;
        const size_t n = msg.size();
        // room first (tokio's bounded mpsc: a permit, then the value)
        size_t q = ch.queued_requests.load();
        while (true) {
            if (ch.closed.load()) return false;
            if (q == 0 || q + n <= ch.buffer_size) {
                if (ch.queued_requests.compare_exchange_weak(q, q + n)) break;
                continue;
            }
            std::unique_lock<std::mutex> lk(ch.mu);
            ch.senders_waiting.fetch_add(1);
            ch.not_full.wait(lk, [&] {
                q = ch.queued_requests.load();
                return ch.closed.load() || q == 0 || q + n <= ch.buffer_size;
            });
            ch.senders_waiting.fetch_sub(1);
        }
        {
            detail::Shard& sh = ch.shard[shard_];
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.queue.push_back(std::move(msg));
        }
        // the actor only ever sleeps on an empty channel, and says so first: either it sees the request counted above, or
        // this load sees its flag (both sequentially consistent)
        if (ch.actor_sleeping.load()) {
            std::lock_guard<std::mutex> lk(ch.mu);
            ch.not_empty.notify_one();
        }
        return true;
    }
    void retain() {
        if (!ch_) return;
        shard_ = ch_->next_shard.fetch_add(1) % detail::SHARDS;
        std::lock_guard<std::mutex> lk(ch_->mu);
        ++ch_->senders;
    }
    void release() {
        if (!ch_) return;
        bool last;
        {
            std::lock_guard<std::mutex> lk(ch_->mu);
            last = --ch_->senders == 0;
            if (last) ch_->closed.store(true); // all senders dropped: rx.recv() returns None (actor.rs:222)
        }
        if (last) {
            ch_->not_empty.notify_all();
            ch_->not_full.notify_all();
            if (ch_->actor.joinable()) ch_->actor.join();
        }
        ch_.reset();
    }
    std::shared_ptr<detail::Channel> ch_;
    uint32_t shard_ = 0;
};

// actor.rs:88-168 (spawn_periodic / spawn_probabilistic / spawn_adaptive differ only in the
// store's cleanup cadence, which never changes a decision; there is one GPU store)
// The loop is written against what it needs from the limiter -- `FLIGHTS`, `submit_batch(std::vector<Request>)`,
// `collect_batch()` -- so that the channel and pipelining logic can be tested without a GPU
// (tests/cpp/test_actor_logic.cpp runs it over a stand-in limiter); RateLimiterActor below is the real one.
template <class Limiter>
class BasicRateLimiterActor {
  public:
    // buffer_size: channel capacity in requests (the server's --buffer-size, config.rs:311, default 100 000)
    // max_batch:   most requests taken per loop turn (<= the store's max_batch)
    // linger:      after the first message, wait up to this long for the queue to reach
    //              min_batch before draining (0 = drain whatever is there: lowest latency)
    static RateLimiterHandle spawn(size_t buffer_size, std::shared_ptr<Limiter> limiter, size_t max_batch = 1 << 16,
                                   std::chrono::microseconds linger = std::chrono::microseconds(0), size_t min_batch = 1) {
        auto ch = std::make_shared<detail::Channel>();
        ch->buffer_size = buffer_size ? buffer_size : 1;
        detail::Channel* raw = ch.get();
        ch->actor = std::thread([raw, limiter, max_batch, linger, min_batch] { run_actor(*raw, *limiter, max_batch, linger, min_batch); });
        return RateLimiterHandle(std::move(ch));
    }

  private:
    static uint64_t clock_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // actor.rs:217-236, draining the queue instead of taking one message -- and pipelined: a drained batch
    // is only SUBMITTED (RateLimiter::submit_batch: marshalled into pinned buffers, transfers and evaluation
    // enqueued); while the GPU works on it the loop drains and marshals the next one, and the replies of
    // the oldest batch in flight go out when the pipeline is full or the queue has nothing to add.
    // Evaluation order == queue order, as with one batch at a time.
    static void run_actor(detail::Channel& ch, Limiter& limiter, size_t max_batch, std::chrono::microseconds linger,
                          size_t min_batch) {
        std::deque<std::deque<RateLimiterMessage>> flying; // submitted batches, oldest first
        size_t first_shard = 0; // the drain starts one shard further every turn
        while (true) {
            std::deque<RateLimiterMessage> msgs;
            if (flying.empty()) { // nothing to answer meanwhile: wait for work
                if (ch.queued_requests.load() == 0) {
                    std::unique_lock<std::mutex> lk(ch.mu);
                    ch.actor_sleeping.store(true);
                    ch.not_empty.wait(lk, [&] { return ch.closed.load() || ch.queued_requests.load() != 0; });
                    ch.actor_sleeping.store(false);
                }
                if (ch.queued_requests.load() == 0 && ch.closed.load()) break; // closed and drained
                if (linger.count() > 0 && ch.queued_requests.load() < min_batch && !ch.closed.load()) {
                    // (senders only signal a sleeping actor: poll in slices of the linger time)
                    const auto until = std::chrono::steady_clock::now() + linger;
                    std::unique_lock<std::mutex> lk(ch.mu);
                    while (!ch.closed.load() && ch.queued_requests.load() < min_batch && std::chrono::steady_clock::now() < until)
                        ch.not_empty.wait_for(lk, linger / 8 + std::chrono::microseconds(1));
                }
            }
            size_t take = 0; // requests
            bool limit = false;
            for (size_t k = 0; k < detail::SHARDS && !limit; ++k) {
                detail::Shard& sh = ch.shard[(first_shard + k) % detail::SHARDS];
                std::lock_guard<std::mutex> lk(sh.mu);
                if (sh.queue.empty()) continue;
                if (msgs.empty() && ch.queued_requests.load() <= max_batch) { // O(1) under the lock the senders need
                    // (queued_requests >= what the shards hold: everything there is fits)
                    msgs.swap(sh.queue);
                    for (const RateLimiterMessage& m : msgs) take += m.size();
                    continue;
                }
                while (!sh.queue.empty()) {
                    if (take != 0 && take + sh.queue.front().size() > max_batch) {
                        limit = true;
                        break;
                    }
                    take += sh.queue.front().size();
                    msgs.push_back(std::move(sh.queue.front()));
                    sh.queue.pop_front();
                }
            }
            first_shard = (first_shard + 1) % detail::SHARDS;
            if (take) {
                ch.queued_requests.fetch_sub(take);
                std::lock_guard<std::mutex> lk(ch.mu);
                ch.batches += 1;
                ch.requests += take;
                if (take > ch.largest_batch) ch.largest_batch = take;
                if (ch.senders_waiting.load() != 0) ch.not_full.notify_all();
            } else if (flying.empty()) {
                std::this_thread::yield(); // (a sender has counted its request and is about to push it)
            }
            const bool took = !msgs.empty();
            if (took) {
                const uint64_t t0 = clock_ns();
                if (submit_throttle_batch(limiter, msgs)) flying.push_back(std::move(msgs));
                ch.ns_submit += clock_ns() - t0;
            }
            // answer the oldest batch when the pipeline is full or the queue had nothing to add
            if (!flying.empty() && (flying.size() >= Limiter::FLIGHTS || !took)) {
                const uint64_t t0 = clock_ns();
                answer_throttle_batch(limiter, flying.front(), &ch.ns_collect);
                const uint64_t t1 = clock_ns();
                flying.pop_front();
                ch.ns_answer += t1 - t0;
                ch.ns_release += clock_ns() - t1;
            }
        }
    }

    static std::vector<Request> requests_of(const std::deque<RateLimiterMessage>& msgs) {
        size_t n = 0;
        for (const RateLimiterMessage& m : msgs) n += m.size();
        std::vector<Request> reqs;
        reqs.reserve(n);
        auto add = [&](const ThrottleRequest& r) {
            reqs.push_back(Request{r.key, r.max_burst, r.count_per_period, r.period, r.quantity, r.timestamp});
        };
        for (const RateLimiterMessage& m : msgs) {
            if (!m.many) add(m.request);
            else
                for (const ThrottleRequest& r : m.many->requests) add(r);
        }
        return reqs;
    }
    // the same reply to every request of a batch (errors)
    static void answer_all(std::deque<RateLimiterMessage>& msgs, const std::string& err) {
        for (RateLimiterMessage& m : msgs) {
            if (!m.many) m.response_tx.set_value(err);
            else m.many->tx.set_value(std::vector<Result<ThrottleResponse>>(m.many->requests.size(), err));
        }
    }

    // actor.rs:238-255 for a whole batch, first half: hand the batch to the limiter.  false: it could not be
    // submitted and every request has been answered with the error.
    static bool submit_throttle_batch(Limiter& limiter, std::deque<RateLimiterMessage>& msgs) {
        try {
            limiter.submit_batch(requests_of(msgs));
            return true;
        } catch (const std::exception& ex) {
            answer_all(msgs, std::string("Rate limit check failed: internal error: ") + ex.what());
            return false;
        }
    }

    // second half: the replies; send errors are ignored like in the reference (the receiver may have given
    // up, actor.rs:229-230)
    static void answer_throttle_batch(Limiter& limiter, std::deque<RateLimiterMessage>& msgs, std::atomic<uint64_t>* ns_collect) {
        std::vector<RateLimitOutcome> out;
        try {
            const uint64_t t0 = clock_ns();
            out = limiter.collect_batch();
            *ns_collect += clock_ns() - t0;
        } catch (const std::exception& ex) {
            answer_all(msgs, std::string("Rate limit check failed: internal error: ") + ex.what());
            return;
        }
        auto reply = [](const RateLimitOutcome& o) -> Result<ThrottleResponse> {
            if (throttlecrab::is_ok(o)) {
                const auto& ok = std::get<0>(o);
                return ThrottleResponse::from(ok.first, ok.second);
            }
            return "Rate limit check failed: " + std::get<1>(o).to_string();
        };
        size_t at = 0;
        for (RateLimiterMessage& m : msgs) {
            if (!m.many) {
                m.response_tx.set_value(reply(out[at++]));
            } else {
                const size_t k = m.many->requests.size();
                std::vector<Result<ThrottleResponse>> rs;
                rs.reserve(k);
                for (size_t i = 0; i < k; ++i) { // (in place: a temporary variant would be moved through its visitor)
                    const RateLimitOutcome& o = out[at++];
                    if (const auto* ok = std::get_if<0>(&o)) rs.emplace_back(std::in_place_index<0>, ThrottleResponse::from(ok->first, ok->second));
                    else rs.push_back(reply(o));
                }
                m.many->tx.set_value(std::move(rs));
            }
        }
    }
};

// actor.rs:88-168 with the GPU store
class RateLimiterActor : public BasicRateLimiterActor<RateLimiter> {
  public:
    static RateLimiterHandle spawn_gpu(size_t buffer_size, GpuStore store, size_t max_batch = 1 << 16,
                                       std::chrono::microseconds linger = std::chrono::microseconds(0), size_t min_batch = 1) {
        return spawn(buffer_size, std::make_shared<RateLimiter>(std::move(store)), max_batch, linger, min_batch);
    }
};

} // namespace server
} // namespace throttlecrab

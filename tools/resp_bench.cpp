// resp_bench.cpp -- RESP bytes in, RESP bytes out: the reference's redis transport loop (transport/redis/mod.rs:126-147: parse a command,
// await the actor, write the reply) without the sockets.  C connection threads each hold a buffer of P pipelined
// `THROTTLE user:<id> 100 1000 3600` commands (what redis-benchmark -P sends); per buffer: resp::Pipeline::parse, run_via_actor (the
// buffer's THROTTLEs travel as one throttle_many message, the actor drains every connection's into one engine batch), replies serialised.
// The reference's figure for this layer: 185 k requests/s through its fastest transport, sockets included.
// usage: resp_bench [connections] [commands per buffer] [buffers per connection]
// build: g++ -O2 -std=c++17 -pthread -Iinclude tools/resp_bench.cpp -Lthrottlecrab_amd -ltcgpu -Wl,-rpath,$PWD/throttlecrab_amd -o tools/bin/resp_bench
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "throttlecrab_actor.hpp"
#include "throttlecrab_resp.hpp"

using namespace throttlecrab;
using namespace throttlecrab::server;

int main(int argc, char** argv) {
    const int conns = argc > 1 ? atoi(argv[1]) : 16;
    const int per_buf = argc > 2 ? atoi(argv[2]) : 1024;
    const int bufs = argc > 3 ? atoi(argv[3]) : 400;
    const uint64_t keys = 1000000;
    const size_t max_batch = 1 << 18;
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(1 << 20, GpuStore(2 * keys, max_batch), max_batch);
    std::atomic<uint64_t> bytes_in{0}, bytes_out{0}, replies{0}, allowed{0};
    auto run = [&](int n_bufs) {
        std::vector<std::thread> th;
        for (int c = 0; c < conns; ++c)
            th.emplace_back([&, c] {
                RateLimiterHandle h = handle;
                uint64_t x = 88172645463325252ULL + c;
                int64_t now = 1700000000LL * 1000000000LL;
                std::string in, out;
                uint64_t bi = 0, bo = 0, rp = 0, al = 0;
                for (int b = 0; b < n_bufs; ++b) {
                    in.clear();
                    for (int i = 0; i < per_buf; ++i) { // (the client's side: not what is measured, but it is in the loop)
                        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                        const std::string key = "user:" + std::to_string(x % keys);
                        in += "*5\r\n$8\r\nTHROTTLE\r\n$" + std::to_string(key.size()) + "\r\n" + key + "\r\n$3\r\n100\r\n$4\r\n1000\r\n$4\r\n3600\r\n";
                    }
                    resp::Pipeline p;
                    const size_t used = p.parse(reinterpret_cast<const uint8_t*>(in.data()), in.size(), [&] { return now += 1000; });
                    out.clear();
                    p.run_via_actor(h, out);
                    if (used != in.size() || p.throttles() != (size_t)per_buf) std::abort();
                    bi += in.size(), bo += out.size();
                    for (size_t at = 0; (at = out.find("*5\r\n:", at)) != std::string::npos; at += 5) ++rp, al += out[at + 5] == '1';
                }
                bytes_in += bi, bytes_out += bo, replies += rp, allowed += al;
            });
        for (auto& t : th) t.join();
    };
    run(bufs / 10 + 1); // warm-up: key inserts, stream probing, pinned buffers
    bytes_in = bytes_out = replies = allowed = 0;
    const auto a = std::chrono::steady_clock::now();
    run(bufs);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    const double n = (double)conns * bufs * per_buf;
    const auto [batches, requests, largest] = handle.drain_stats();
    std::printf("resp: %d connections x %d buffers of %d THROTTLE commands: %.2f M commands/s (%.0f MB/s in, %.0f MB/s out; %llu replies, %.1f %% allowed; "
                "engine batches avg %.0f, largest %llu)\n",
                conns, bufs, per_buf, n / dt / 1e6, bytes_in.load() / dt / 1e6, bytes_out.load() / dt / 1e6, (unsigned long long)replies.load(),
                100.0 * allowed.load() / (double)replies.load(), (double)requests / (double)(batches ? batches : 1), (unsigned long long)largest);
    return replies.load() == (uint64_t)n ? 0 : 1;
}

// throttlecrab_metrics.hpp -- Metrics of the reference server over the engine's device counters
// (C++17, header only): SURVEY.md section 8(f) row 3.
//
// Mirrors throttlecrab-server/src/metrics.rs:
//   Metrics / MetricsBuilder (max_denied_keys, 0 = disabled, capped at 10 000)   :79-148
//   record_request / record_request_with_key / record_error                        :162-214
//   escape_prometheus_label                                                        :217-233
//   export_prometheus (exact text, metric names, blank lines, label order)         :236-311
// What differs: the reference's transports bump allowed/denied and a capped HashMap of denied
// keys after every reply (metrics.rs:24-76).  Here the decisions are counted where they are
// made: the evaluation kernels keep allowed/denied/error totals (tc_counters) and an exact
// denial counter per key (TC_CFG_TRACK_DENIED, tc_top_denied); the transports only count
// requests per transport.  `Metrics::snapshot_from_engine` folds the device numbers in; with
// several GPUs the per-GPU counter blocks are all-gathered first (bench.py, sharded.py).
#pragma once

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "tcgpu.h"

namespace throttlecrab {
namespace server {

enum class Transport { Http, Grpc, Redis }; // metrics.rs:315-320

class Metrics {
  public:
    static constexpr size_t MAX_KEY_LENGTH = 256;          // metrics.rs:12
    static constexpr size_t MAX_DENIED_KEYS_LIMIT = 10000; // metrics.rs:17

    // MetricsBuilder::max_denied_keys (default 100, 0 disables, capped): metrics.rs:103-121
    explicit Metrics(size_t max_denied_keys = 100)
        : start_(std::chrono::steady_clock::now()),
          max_denied_keys_(max_denied_keys > MAX_DENIED_KEYS_LIMIT ? MAX_DENIED_KEYS_LIMIT : max_denied_keys) {}

    std::atomic<uint64_t> total_requests{0}, http_requests{0}, grpc_requests{0}, redis_requests{0};
    std::atomic<uint64_t> requests_allowed{0}, requests_denied{0}, requests_errors{0};

    // metrics.rs:176-193 (host-side recording, for hosts that do not take the device counters)
    void record_request(Transport t, bool allowed) {
        total_requests.fetch_add(1, std::memory_order_relaxed);
        by_transport(t).fetch_add(1, std::memory_order_relaxed);
        (allowed ? requests_allowed : requests_denied).fetch_add(1, std::memory_order_relaxed);
    }
    // metrics.rs:196-206
    void record_error(Transport t) {
        total_requests.fetch_add(1, std::memory_order_relaxed);
        requests_errors.fetch_add(1, std::memory_order_relaxed);
        by_transport(t).fetch_add(1, std::memory_order_relaxed);
    }
    // transports that leave the decision counters to the device only say "n requests came in here"
    void record_transport(Transport t, uint64_t n) { by_transport(t).fetch_add(n, std::memory_order_relaxed); }

    // Replace the decision counters and the top-denied list by what the engine counted.
    // (total = allowed + denied + errors, like metrics.rs's counter consistency test :384-411)
    int snapshot_from_engine(tc_engine* e) {
        uint64_t c[TC_CNT_COUNT];
        int rc = tc_counters(e, c);
        if (rc != TC_E_OK) return rc;
        requests_allowed.store(c[TC_CNT_ALLOWED]);
        requests_denied.store(c[TC_CNT_DENIED]);
        requests_errors.store(c[TC_CNT_ERRORS]);
        total_requests.store(c[TC_CNT_TOTAL]);
        top_.clear();
        if (max_denied_keys_ == 0) return TC_E_OK;
        std::vector<uint64_t> counts(max_denied_keys_);
        uint32_t n = 0;
        // string keys: by KEY, the keys a sweep has unbound included (the reference counts by key, metrics.rs:24-76)
        {
            std::vector<uint32_t> off(max_denied_keys_ + 1, 0);
            std::vector<uint8_t> bytes(max_denied_keys_ * 64 + 64);
            while ((rc = tc_top_denied_keys(e, (uint32_t)max_denied_keys_, bytes.data(), bytes.size(), off.data(), counts.data(), &n)) ==
                       TC_E_INVALID_ARG && bytes.size() < (1u << 30))
                bytes.resize(bytes.size() * 8);
            if (rc == TC_E_OK) {
                for (uint32_t i = 0; i < n; ++i) {
                    const size_t len = off[i + 1] - off[i];
                    if (len > MAX_KEY_LENGTH) continue; // metrics.rs:39-42: over-long keys are not tracked
                    top_.emplace_back(std::string((const char*)bytes.data() + off[i], len), counts[i]);
                }
                return TC_E_OK;
            }
            if (rc != TC_E_UNSUPPORTED) return rc;
        }
        // slot-mode engine (label = the slot id), or no TC_CFG_TRACK_DENIED (counters only)
        std::vector<uint32_t> slots(max_denied_keys_);
        rc = tc_top_denied(e, (uint32_t)max_denied_keys_, slots.data(), counts.data(), &n);
        if (rc == TC_E_UNSUPPORTED) return TC_E_OK;
        if (rc != TC_E_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) top_.emplace_back("slot:" + std::to_string(slots[i]), counts[i]);
        return TC_E_OK;
    }
    // for hosts that track denied keys themselves: most denied first
    void set_top_denied(std::vector<std::pair<std::string, uint64_t>> top) {
        if (top.size() > max_denied_keys_) top.resize(max_denied_keys_);
        top_ = std::move(top);
    }

    uint64_t uptime_seconds() const {
        return (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - start_).count();
    }

    // metrics.rs:217-233 (`c.is_control()`: C0, DEL and C1 controls; `c as u8` truncates)
    static std::string escape_prometheus_label(const std::string& s) {
        std::string out;
        out.reserve(s.size() * 2);
        for (size_t i = 0; i < s.size();) {
            const unsigned char c = (unsigned char)s[i];
            if (c == '"') out += "\\\"", ++i;
            else if (c == '\\') out += "\\\\", ++i;
            else if (c == '\n') out += "\\n", ++i;
            else if (c == '\r') out += "\\r", ++i;
            else if (c == '\t') out += "\\t", ++i;
            else if (c < 0x20 || c == 0x7F) {
                char b[8];
                std::snprintf(b, sizeof b, "\\x%02x", c);
                out += b;
                ++i;
            } else if (c == 0xC2 && i + 1 < s.size() && (unsigned char)s[i + 1] >= 0x80 && (unsigned char)s[i + 1] <= 0x9F) {
                char b[8]; // U+0080..U+009F: control; `c as u8` is the code point itself
                std::snprintf(b, sizeof b, "\\x%02x", (unsigned char)s[i + 1]);
                out += b;
                i += 2;
            } else {
                out += (char)c;
                ++i;
            }
        }
        return out;
    }

    // metrics.rs:236-311, byte for byte
    std::string export_prometheus() const {
        std::string o;
        o.reserve(600);
        o += "# HELP throttlecrab_uptime_seconds Time since server start in seconds\n";
        o += "# TYPE throttlecrab_uptime_seconds gauge\n";
        o += "throttlecrab_uptime_seconds " + std::to_string(uptime_seconds()) + "\n\n";
        o += "# HELP throttlecrab_requests_total Total number of requests processed\n";
        o += "# TYPE throttlecrab_requests_total counter\n";
        o += "throttlecrab_requests_total " + std::to_string(total_requests.load()) + "\n\n";
        o += "# HELP throttlecrab_requests_by_transport Total requests by transport type\n";
        o += "# TYPE throttlecrab_requests_by_transport counter\n";
        o += "throttlecrab_requests_by_transport{transport=\"http\"} " + std::to_string(http_requests.load()) + "\n";
        o += "throttlecrab_requests_by_transport{transport=\"grpc\"} " + std::to_string(grpc_requests.load()) + "\n";
        o += "throttlecrab_requests_by_transport{transport=\"redis\"} " + std::to_string(redis_requests.load()) + "\n\n";
        o += "# HELP throttlecrab_requests_allowed Total requests allowed\n";
        o += "# TYPE throttlecrab_requests_allowed counter\n";
        o += "throttlecrab_requests_allowed " + std::to_string(requests_allowed.load()) + "\n\n";
        o += "# HELP throttlecrab_requests_denied Total requests denied\n";
        o += "# TYPE throttlecrab_requests_denied counter\n";
        o += "throttlecrab_requests_denied " + std::to_string(requests_denied.load()) + "\n\n";
        o += "# HELP throttlecrab_requests_errors Total internal errors\n";
        o += "# TYPE throttlecrab_requests_errors counter\n";
        o += "throttlecrab_requests_errors " + std::to_string(requests_errors.load()) + "\n\n";
        if (max_denied_keys_ != 0) {
            o += "# HELP throttlecrab_top_denied_keys Top keys by denial count\n";
            o += "# TYPE throttlecrab_top_denied_keys gauge\n";
            for (size_t rank = 0; rank < top_.size(); ++rank)
                o += "throttlecrab_top_denied_keys{key=\"" + escape_prometheus_label(top_[rank].first) + "\",rank=\"" +
                     std::to_string(rank + 1) + "\"} " + std::to_string(top_[rank].second) + "\n";
        }
        return o;
    }

  private:
    std::atomic<uint64_t>& by_transport(Transport t) {
        return t == Transport::Http ? http_requests : t == Transport::Grpc ? grpc_requests : redis_requests;
    }
    std::chrono::steady_clock::time_point start_;
    size_t max_denied_keys_;
    std::vector<std::pair<std::string, uint64_t>> top_;
};

} // namespace server
} // namespace throttlecrab

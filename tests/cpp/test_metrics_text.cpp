// Prometheus text of include/throttlecrab_metrics.hpp vs throttlecrab-server/src/metrics.rs:236-311
// and the assertions of metrics.rs:327-412 / tests/metrics_test.rs / tests/denied_keys_test.rs (no GPU).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "throttlecrab_metrics.hpp"

using namespace throttlecrab::server;

#define CHECK(c)                                                                \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

static bool has(const std::string& s, const char* needle) { return s.find(needle) != std::string::npos; }

int main() {
    { // metrics.rs:342-361, 364-381
        Metrics m;
        m.record_request(Transport::Http, true);
        m.record_request(Transport::Grpc, false);
        CHECK(m.total_requests == 2 && m.http_requests == 1 && m.grpc_requests == 1 && m.requests_allowed == 1 && m.requests_denied == 1);
        const std::string o = m.export_prometheus();
        CHECK(has(o, "throttlecrab_uptime_seconds") && has(o, "throttlecrab_requests_total 2") &&
              has(o, "throttlecrab_requests_allowed 1") && has(o, "throttlecrab_requests_denied 1") &&
              has(o, "throttlecrab_requests_by_transport{transport=\"http\"} 1") &&
              has(o, "throttlecrab_requests_by_transport{transport=\"grpc\"} 1"));
    }
    { // metrics.rs:384-411
        Metrics m;
        m.record_request(Transport::Http, true);
        m.record_request(Transport::Http, false);
        m.record_request(Transport::Grpc, true);
        m.record_request(Transport::Grpc, false);
        m.record_error(Transport::Http);
        CHECK(m.total_requests == 5 && m.http_requests + m.grpc_requests == 5);
        CHECK(m.requests_allowed + m.requests_denied + m.requests_errors == 5 && m.requests_errors == 1);
    }
    { // exact text, tracking on with a list and off (denied_keys_test.rs:89-107)
        Metrics m(3);
        m.record_transport(Transport::Redis, 7);
        m.total_requests = 7;
        m.requests_allowed = 4;
        m.requests_denied = 3;
        m.set_top_denied({{"user:123", 2}, {"we\"ird\\key\n\x01", 1}, {"c", 1}, {"dropped: beyond max", 1}});
        std::string o = m.export_prometheus();
        const size_t up = o.find("\nthrottlecrab_uptime_seconds ") + 1;
        o.replace(up, o.find('\n', up) - up, "throttlecrab_uptime_seconds 0");
        const std::string want =
            "# HELP throttlecrab_uptime_seconds Time since server start in seconds\n"
            "# TYPE throttlecrab_uptime_seconds gauge\n"
            "throttlecrab_uptime_seconds 0\n\n"
            "# HELP throttlecrab_requests_total Total number of requests processed\n"
            "# TYPE throttlecrab_requests_total counter\n"
            "throttlecrab_requests_total 7\n\n"
            "# HELP throttlecrab_requests_by_transport Total requests by transport type\n"
            "# TYPE throttlecrab_requests_by_transport counter\n"
            "throttlecrab_requests_by_transport{transport=\"http\"} 0\n"
            "throttlecrab_requests_by_transport{transport=\"grpc\"} 0\n"
            "throttlecrab_requests_by_transport{transport=\"redis\"} 7\n\n"
            "# HELP throttlecrab_requests_allowed Total requests allowed\n"
            "# TYPE throttlecrab_requests_allowed counter\n"
            "throttlecrab_requests_allowed 4\n\n"
            "# HELP throttlecrab_requests_denied Total requests denied\n"
            "# TYPE throttlecrab_requests_denied counter\n"
            "throttlecrab_requests_denied 3\n\n"
            "# HELP throttlecrab_requests_errors Total internal errors\n"
            "# TYPE throttlecrab_requests_errors counter\n"
            "throttlecrab_requests_errors 0\n\n"
            "# HELP throttlecrab_top_denied_keys Top keys by denial count\n"
            "# TYPE throttlecrab_top_denied_keys gauge\n"
            "throttlecrab_top_denied_keys{key=\"user:123\",rank=\"1\"} 2\n"
            "throttlecrab_top_denied_keys{key=\"we\\\"ird\\\\key\\n\\x01\",rank=\"2\"} 1\n"
            "throttlecrab_top_denied_keys{key=\"c\",rank=\"3\"} 1\n";
        if (o != want) std::fprintf(stderr, "got:\n%s\nwant:\n%s\n", o.c_str(), want.c_str());
        CHECK(o == want);
        Metrics off(0);
        off.requests_denied = 100;
        const std::string o2 = off.export_prometheus();
        CHECK(!has(o2, "throttlecrab_top_denied_keys") && has(o2, "throttlecrab_requests_total") &&
              has(o2, "throttlecrab_requests_denied 100"));
        Metrics capped(50000);
        std::vector<std::pair<std::string, uint64_t>> many;
        for (int i = 0; i < 20000; ++i) many.emplace_back("user:" + std::to_string(i), 1);
        capped.set_top_denied(many);
        const std::string o3 = capped.export_prometheus();
        size_t lines = 0;
        for (size_t p = 0; (p = o3.find("throttlecrab_top_denied_keys{", p)) != std::string::npos; ++p) ++lines;
        CHECK(lines == 10000); // MAX_DENIED_KEYS_LIMIT (denied_keys_test.rs:70-86)
    }
    std::puts("all tests passed");
    return 0;
}

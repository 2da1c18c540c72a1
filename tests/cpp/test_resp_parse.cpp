// RESP front-end parsing tests (no GPU): protocol handling and argument validation written after
// throttlecrab-server/src/transport/redis_test.rs, redis_security_test.rs and redis/resp.rs tests.
// build: g++ -std=c++17 -Iinclude tests/cpp/test_resp_parse.cpp -o tests/cpp/test_resp_parse
#include <cstdio>
#include <cstdlib>
#include <string>

#include "throttlecrab_resp.hpp"

using namespace throttlecrab::server::resp;

#define CHECK(c)                                                                \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

static std::string cmd(std::initializer_list<std::string> args) {
    std::string s = "*" + std::to_string(args.size()) + "\r\n";
    for (const std::string& a : args) s += "$" + std::to_string(a.size()) + "\r\n" + a + "\r\n";
    return s;
}

static Pipeline parse_all(const std::string& wire, size_t* consumed = nullptr) {
    Pipeline p;
    int64_t t = 1700000000LL * 1000000000LL;
    size_t c = p.parse((const uint8_t*)wire.data(), wire.size(), [&] { return t++; });
    if (consumed) *consumed = c;
    return p;
}

int main() {
    // parse_i64 == str::parse::<i64>
    int64_t v;
    CHECK(parse_i64("0", &v) && v == 0);
    CHECK(parse_i64("+17", &v) && v == 17);
    CHECK(parse_i64("-9223372036854775808", &v) && v == INT64_MIN);
    CHECK(parse_i64("9223372036854775807", &v) && v == INT64_MAX);
    CHECK(!parse_i64("9223372036854775808", &v));
    CHECK(!parse_i64("", &v) && !parse_i64("-", &v) && !parse_i64(" 1", &v) && !parse_i64("1.5", &v) && !parse_i64("abc", &v));

    // PING / PING msg / PING a b   (redis_test.rs ping tests; mod.rs:210-220)
    {
        Pipeline p = parse_all(cmd({"PING"}) + cmd({"ping", "hello"}) + cmd({"PING", "a", "b"}));
        CHECK(p.commands() == 3 && p.throttles() == 0);
        CHECK(p.immediate_reply(0) == "+PONG\r\n");
        CHECK(p.immediate_reply(1) == "$5\r\nhello\r\n");
        CHECK(p.immediate_reply(2) == "-ERR wrong number of arguments for 'ping' command\r\n");
    }
    // THROTTLE argument validation (mod.rs:222-262)
    {
        Pipeline p = parse_all(cmd({"THROTTLE", "k"}) + cmd({"THROTTLE", "k", "x", "1", "1"}) + cmd({"THROTTLE", "k", "1", "y", "1"}) +
                               cmd({"THROTTLE", "k", "1", "1", "z"}) + cmd({"THROTTLE", "k", "1", "1", "1", "q"}) +
                               cmd({"THROTTLE", "k", "1", "1", "1", "1", "1"}) + cmd({"throttle", "k", "10", "100", "60"}) +
                               cmd({"THROTTLE", "k2", "10", "100", "60", "5"}) + cmd({"FLUSHALL"}) + std::string("*0\r\n") +
                               std::string("+OK\r\n") + std::string("*1\r\n:5\r\n") + std::string("*5\r\n$8\r\nTHROTTLE\r\n$-1\r\n$1\r\n1\r\n$1\r\n1\r\n$1\r\n1\r\n") +
                               std::string("*5\r\n$8\r\nTHROTTLE\r\n$1\r\nk\r\n:7\r\n:70\r\n:60\r\n"));
        CHECK(p.protocol_error.empty());
        CHECK(p.commands() == 14 && p.throttles() == 3);
        CHECK(p.immediate_reply(0) == "-ERR wrong number of arguments for 'throttle' command\r\n");
        CHECK(p.immediate_reply(1) == "-ERR invalid max_burst\r\n");
        CHECK(p.immediate_reply(2) == "-ERR invalid count_per_period\r\n");
        CHECK(p.immediate_reply(3) == "-ERR invalid period\r\n");
        CHECK(p.immediate_reply(4) == "-ERR invalid quantity\r\n");
        CHECK(p.immediate_reply(5) == "-ERR wrong number of arguments for 'throttle' command\r\n");
        CHECK(p.is_throttle(6) && p.is_throttle(7));
        CHECK(p.immediate_reply(8) == "-ERR unknown command 'FLUSHALL'\r\n");
        CHECK(p.immediate_reply(9) == "-ERR empty command\r\n");
        CHECK(p.immediate_reply(10) == "-ERR expected array of commands\r\n");
        CHECK(p.immediate_reply(11) == "-ERR invalid command format\r\n");
        CHECK(p.immediate_reply(12) == "-ERR invalid key\r\n");
        CHECK(p.is_throttle(13)); // integer-typed arguments are accepted (mod.rs:292)
        CHECK(p.max_burst[0] == 10 && p.count_per_period[0] == 100 && p.period[0] == 60 && p.quantity[0] == 1);
        CHECK(p.quantity[1] == 5 && p.max_burst[2] == 7 && p.count_per_period[2] == 70);
        CHECK(p.key_off.size() == 4 && p.key_off[1] == 1 && p.key_off[2] == 3 && p.key_off[3] == 4);
        CHECK(std::string(p.key_bytes.begin(), p.key_bytes.end()) == "kk2k");
        CHECK(p.now_ns[0] < p.now_ns[1] && p.now_ns[1] < p.now_ns[2]); // one clock reading per THROTTLE
    }
    // incomplete input is left in the buffer; QUIT stops the pipeline (mod.rs:126-147)
    {
        const std::string full = cmd({"THROTTLE", "abc", "5", "10", "60"});
        for (size_t cut = 0; cut < full.size(); ++cut) {
            size_t consumed = 99;
            Pipeline p = parse_all(full.substr(0, cut), &consumed);
            CHECK(consumed == 0 && p.commands() == 0 && p.protocol_error.empty());
        }
        size_t consumed = 0;
        const std::string two = cmd({"PING"}) + cmd({"QUIT"}) + cmd({"PING"});
        Pipeline p = parse_all(two, &consumed);
        CHECK(p.quit && p.commands() == 2 && p.immediate_reply(1) == "+OK\r\n");
        CHECK(consumed == cmd({"PING"}).size() + cmd({"QUIT"}).size());
    }
    // hardening (redis_security_test.rs; resp.rs:8-10,99,120,138)
    {
        CHECK(!parse_all("$536870913\r\n").protocol_error.empty());                 // > 512 MB
        CHECK(!parse_all("$-2\r\n").protocol_error.empty());
        CHECK(!parse_all("*1048577\r\n").protocol_error.empty());                   // > 1 M elements
        CHECK(!parse_all("*-5\r\n").protocol_error.empty());
        CHECK(!parse_all("?what\r\n").protocol_error.empty());                      // bad type marker
        CHECK(!parse_all(":12x\r\n").protocol_error.empty());
        CHECK(!parse_all("$2\r\n\xff\xfe\r\n").protocol_error.empty());             // not UTF-8
        std::string deep;
        for (int i = 0; i < 129; ++i) deep += "*1\r\n";
        deep += ":1\r\n";
        CHECK(parse_all(deep).protocol_error == "Maximum array nesting depth exceeded");
        std::string ok_deep;
        for (int i = 0; i < 128; ++i) ok_deep += "*1\r\n";
        ok_deep += ":1\r\n";
        Pipeline p = parse_all(ok_deep);
        CHECK(p.protocol_error.empty() && p.commands() == 1 && p.immediate_reply(0) == "-ERR invalid command format\r\n");
        // unicode and odd keys go through untouched (store_test_suite.rs:289-338)
        Pipeline u = parse_all(cmd({"THROTTLE", "\xF0\x9F\xA6\x80\xF0\x9F\x94\xA5", "1", "1", "1"}) + cmd({"THROTTLE", "", "1", "1", "1"}));
        CHECK(u.throttles() == 2 && u.key_off[1] == 8 && u.key_off[2] == 8);
        // PING echoing a null array comes back as the reference re-serialises it
        Pipeline e = parse_all("*2\r\n$4\r\nPING\r\n*-1\r\n");
        CHECK(e.immediate_reply(0) == "*0\r\n");
    }
    std::puts("all tests passed");
    return 0;
}

// throttlecrab_resp.hpp -- RESP `THROTTLE` front end for the batched engine (C++17, header
// only): SURVEY.md section 8(f) row 2.  It turns the bytes of one connection's pipelined
// commands straight into the engine's key arena + request columns, runs every THROTTLE of the
// buffer as ONE batch (in command order), and writes the replies in command order.
//
// Mirrors, with the reference's behaviour and reply strings:
//   RESP parsing + limits        throttlecrab-server/src/transport/redis/resp.rs:8-10,40-177
//   command dispatch             throttlecrab-server/src/transport/redis/mod.rs:150-208
//   PING / THROTTLE / QUIT       throttlecrab-server/src/transport/redis/mod.rs:210-287
//   integer arguments            throttlecrab-server/src/transport/redis/mod.rs:289-295 (str::parse::<i64>)
//   reply serialisation          throttlecrab-server/src/transport/redis/resp.rs:190-230
//   seconds truncation           throttlecrab-server/src/types.rs:87-96
// The reference handles one command at a time per connection (mod.rs:126-147: parse, await the
// actor, write, repeat); doing a whole buffer as one batch is indistinguishable because the
// engine applies a batch in index order.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <string_view>
#include <variant>
#include <vector>

#include "tcgpu.h"

namespace throttlecrab {
namespace server {
namespace resp {

constexpr int64_t MAX_BULK_STRING_SIZE = 512LL * 1024 * 1024; // resp.rs:8
constexpr int64_t MAX_ARRAY_SIZE = 1024 * 1024;               // resp.rs:9
constexpr size_t MAX_ARRAY_DEPTH = 128;                       // resp.rs:10

// Rust's str::parse::<i64>: optional sign, ASCII digits only, no overflow
inline bool parse_i64(std::string_view s, int64_t* out) {
    size_t i = 0;
    bool neg = false;
    if (s.empty()) return false;
    if (s[0] == '+' || s[0] == '-') {
        neg = s[0] == '-';
        i = 1;
    }
    if (i == s.size()) return false;
    // accumulate negatively so that i64::MIN parses
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        const char ch = s[i];
        if (ch < '0' || ch > '9') return false;
        const int d = ch - '0';
        if (__builtin_mul_overflow(v, (int64_t)10, &v)) return false;
        if (__builtin_sub_overflow(v, (int64_t)d, &v)) return false;
    }
    if (!neg) {
        if (v == INT64_MIN) return false;
        v = -v;
    }
    *out = v;
    return true;
}

// str::from_utf8 acceptance (the reference converts every line and bulk string)
inline bool valid_utf8(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t c = p[i];
        if (c < 0x80) {
            ++i;
            continue;
        }
        int extra;
        uint32_t cp;
        if (c >= 0xC2 && c <= 0xDF) extra = 1, cp = c & 0x1F;
        else if (c >= 0xE0 && c <= 0xEF) extra = 2, cp = c & 0x0F;
        else if (c >= 0xF0 && c <= 0xF4) extra = 3, cp = c & 0x07;
        else return false;
        if (i + (size_t)extra >= n) return false;
        for (int k = 1; k <= extra; ++k) {
            const uint8_t cc = p[i + k];
            if ((cc & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (extra == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (extra == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        i += extra + 1;
    }
    return true;
}

enum class Kind : uint8_t { Simple, Error, Integer, Bulk, NullBulk, Array };

// one direct element of a command array (nested arrays are kept as an extent only)
struct Elem {
    Kind kind;
    int64_t integer = 0;     // Integer
    const uint8_t* p = nullptr; // Simple / Error / Bulk: the text; Array: the raw encoding
    size_t n = 0;
};

enum class ParseStatus { Ok, NeedMore, ProtocolError };

// Minimal recursive RESP reader over a byte range.
class Reader {
  public:
    Reader(const uint8_t* data, size_t len) : d_(data), n_(len) {}
    std::string error;

    // Reads one value starting at `at`; on Ok sets *end.  Fills `top` (kind etc.) and, for an
    // array at depth 0, its direct elements.
    ParseStatus value(size_t at, size_t* end, Elem* top, std::vector<Elem>* elems, size_t depth) {
        if (at >= n_) return ParseStatus::NeedMore;
        const uint8_t marker = d_[at];
        size_t line_end, next;
        switch (marker) {
            case '+':
            case '-': {
                if (!line(at, &line_end, &next)) return ParseStatus::NeedMore;
                if (!valid_utf8(d_ + at + 1, line_end - at - 1)) return fail("invalid utf-8 sequence");
                top->kind = marker == '+' ? Kind::Simple : Kind::Error;
                top->p = d_ + at + 1;
                top->n = line_end - at - 1;
                *end = next;
                return ParseStatus::Ok;
            }
            case ':': {
                if (!line(at, &line_end, &next)) return ParseStatus::NeedMore;
                int64_t v;
                if (!parse_i64(sv(at + 1, line_end), &v)) return fail("invalid digit found in string");
                top->kind = Kind::Integer;
                top->integer = v;
                *end = next;
                return ParseStatus::Ok;
            }
            case '$': {
                if (!line(at, &line_end, &next)) return ParseStatus::NeedMore;
                int64_t len;
                if (!parse_i64(sv(at + 1, line_end), &len)) return fail("invalid digit found in string");
                if (len == -1) {
                    top->kind = Kind::NullBulk;
                    *end = next;
                    return ParseStatus::Ok;
                }
                if (len < 0 || len > MAX_BULK_STRING_SIZE) return fail("Invalid bulk string length: " + std::to_string(len));
                if (n_ < next + (size_t)len + 2) return ParseStatus::NeedMore;
                if (!valid_utf8(d_ + next, (size_t)len)) return fail("invalid utf-8 sequence");
                top->kind = Kind::Bulk;
                top->p = d_ + next;
                top->n = (size_t)len;
                *end = next + (size_t)len + 2; // the reference skips the 2 trailer bytes unchecked (resp.rs:112)
                return ParseStatus::Ok;
            }
            case '*': {
                if (depth >= MAX_ARRAY_DEPTH) return fail("Maximum array nesting depth exceeded");
                if (!line(at, &line_end, &next)) return ParseStatus::NeedMore;
                int64_t count;
                if (!parse_i64(sv(at + 1, line_end), &count)) return fail("invalid digit found in string");
                top->kind = Kind::Array;
                top->p = d_ + at;
                if (count == -1) { // null array == empty array (resp.rs:131-134)
                    top->n = next - at;
                    *end = next;
                    return ParseStatus::Ok;
                }
                if (count < 0 || count > MAX_ARRAY_SIZE) return fail("Invalid array size: " + std::to_string(count));
                size_t pos = next;
                for (int64_t i = 0; i < count; ++i) {
                    Elem e;
                    size_t e_end;
                    const ParseStatus st = value(pos, &e_end, &e, nullptr, depth + 1);
                    if (st != ParseStatus::Ok) return st;
                    if (elems) elems->push_back(e);
                    pos = e_end;
                }
                top->n = pos - at;
                *end = pos;
                return ParseStatus::Ok;
            }
            default:
                return fail(std::string("Invalid RESP type marker: ") + (char)marker);
        }
    }

  private:
    ParseStatus fail(std::string msg) {
        error = std::move(msg);
        return ParseStatus::ProtocolError;
    }
    std::string_view sv(size_t a, size_t b) const { return std::string_view((const char*)d_ + a, b - a); }
    // CRLF-terminated line starting at `at` (resp.rs:165-173)
    bool line(size_t at, size_t* line_end, size_t* next) const {
        for (size_t i = at; i + 1 < n_; ++i)
            if (d_[i] == '\r' && d_[i + 1] == '\n') {
                *line_end = i;
                *next = i + 2;
                return true;
            }
        return false;
    }
    const uint8_t* d_;
    size_t n_;
};

inline void put_integer(std::string& out, int64_t v) {
    out += ':';
    out += std::to_string(v);
    out += "\r\n";
}
inline void put_error(std::string& out, const std::string& msg) {
    out += '-';
    out += msg;
    out += "\r\n";
}
inline void put_simple(std::string& out, const char* s) {
    out += '+';
    out += s;
    out += "\r\n";
}
inline void put_elem(std::string& out, const struct Elem& e);
// Re-serialises an already validated raw array the way RespSerializer would after the
// reference parsed it into a RespValue (a null array "*-1" comes back as "*0", ":+5" as ":5").
inline void canonical_array(std::string& out, const uint8_t* raw, size_t n) {
    Reader rd(raw, n);
    Elem top;
    std::vector<Elem> elems;
    size_t end;
    if (rd.value(0, &end, &top, &elems, 0) != ParseStatus::Ok) return; // cannot happen: validated when first read
    out += '*';
    out += std::to_string(elems.size());
    out += "\r\n";
    for (const Elem& e : elems) put_elem(out, e);
}
// RespSerializer::serialize of an arbitrary parsed element (PING echoes its argument, mod.rs:213-216)
inline void put_elem(std::string& out, const Elem& e) {
    switch (e.kind) {
        case Kind::Simple: out += '+'; out.append((const char*)e.p, e.n); out += "\r\n"; break;
        case Kind::Error: out += '-'; out.append((const char*)e.p, e.n); out += "\r\n"; break;
        case Kind::Integer: put_integer(out, e.integer); break;
        case Kind::NullBulk: out += "$-1\r\n"; break;
        case Kind::Bulk:
            out += '$';
            out += std::to_string(e.n);
            out += "\r\n";
            out.append((const char*)e.p, e.n);
            out += "\r\n";
            break;
        case Kind::Array: canonical_array(out, e.p, e.n); break;
    }
}

// All complete commands of one buffer, THROTTLEs gathered into engine-ready columns.
class Pipeline {
  public:
    using Clock = std::function<int64_t()>; // ns since UNIX_EPOCH; called once per THROTTLE (mod.rs:264)

    // Parses as many complete commands as `data` holds.  Returns the bytes consumed; stops
    // after QUIT (the reference closes the connection, mod.rs:143-146).  On a protocol error
    // sets protocol_error (the reference drops the connection, mod.rs:126) and stops.
    size_t parse(const uint8_t* data, size_t len, const Clock& clock) {
        Reader rd(data, len);
        size_t pos = 0;
        while (pos < len && !quit) {
            Elem top;
            std::vector<Elem> args;
            size_t end;
            const ParseStatus st = rd.value(pos, &end, &top, &args, 0);
            if (st == ParseStatus::NeedMore) break;
            if (st == ParseStatus::ProtocolError) {
                protocol_error = rd.error;
                break;
            }
            pos = end;
            command(top, args, clock);
        }
        return pos;
    }

    size_t commands() const { return cmds_.size(); }
    size_t throttles() const { return max_burst.size(); }

    // engine-ready request columns (key arena = key_bytes + key_off[n+1])
    std::vector<uint8_t> key_bytes;
    std::vector<uint32_t> key_off{0};
    std::vector<int64_t> max_burst, count_per_period, period, quantity, now_ns;
    bool quit = false;
    std::string protocol_error;

    // Runs every THROTTLE as one batch and appends all replies, in command order, to `out`.
    // Returns the engine's return code (TC_E_OK, or TC_E_TABLE_FULL: those keys answer with an error).
    int run(tc_engine* e, std::string& out) {
        const size_t n = throttles();
        std::vector<tc_decision> dec(n); // one 32-byte record per THROTTLE; `limit` is the command's max_burst
        int rc = TC_E_OK;
        if (n) {
            if (key_bytes.empty()) key_bytes.push_back(0);
            tc_batch b;
            std::memset(&b, 0, sizeof b);
            b.struct_size = sizeof b;
            b.n = n;
            b.key_bytes = key_bytes.data();
            b.key_off = key_off.data();
            b.max_burst = max_burst.data();
            b.count_per_period = count_per_period.data();
            b.period = period.data();
            b.quantity = quantity.data();
            b.now_ns = now_ns.data();
            b.decisions = dec.data();
            rc = tc_rate_limit_batch_keys(e, &b);
        }
        const bool usable = rc == TC_E_OK || rc == TC_E_TABLE_FULL;
        for (const Cmd& c : cmds_) {
            if (c.throttle < 0) {
                out += c.reply;
                continue;
            }
            const size_t i = (size_t)c.throttle;
            if (!usable) {
                put_error(out, std::string("ERR Rate limit check failed: internal error: ") + tc_last_error(e));
            } else if (dec[i].status == TC_OK) {
                // [allowed, limit, remaining, reset_after s, retry_after s] (mod.rs:274-283, types.rs:87-96)
                out += "*5\r\n";
                put_integer(out, dec[i].allowed ? 1 : 0);
                put_integer(out, max_burst[i]);
                put_integer(out, dec[i].remaining);
                put_integer(out, dec[i].reset_after_ns / 1000000000LL);
                put_integer(out, dec[i].retry_after_ns / 1000000000LL);
            } else if (dec[i].status == TC_NEGATIVE_QUANTITY) { // CellError Display (core/mod.rs:58-66)
                put_error(out, "ERR Rate limit check failed: negative quantity: " + std::to_string(quantity[i]));
            } else if (dec[i].status == TC_INVALID_RATE_LIMIT) {
                put_error(out, "ERR Rate limit check failed: invalid rate limit parameters");
            } else {
                put_error(out, "ERR Rate limit check failed: internal error: outside the validated domain");
            }
        }
        return rc;
    }

    // The same through the batch-draining actor (throttlecrab_actor.hpp: `Handle` = RateLimiterHandle): the
    // buffer's THROTTLEs travel as ONE throttle_many message, so the buffers of many connections end up in
    // one engine batch -- what redis/mod.rs:264-287 does per command with handle.throttle(req).await.
    template <class Handle>
    void run_via_actor(Handle& handle, std::string& out) {
        using Req = typename Handle::request_type;
        const size_t n = throttles();
        std::vector<Req> reqs;
        reqs.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            Req r;
            r.key.assign((const char*)key_bytes.data() + key_off[i], key_off[i + 1] - key_off[i]);
            r.max_burst = max_burst[i];
            r.count_per_period = count_per_period[i];
            r.period = period[i];
            r.quantity = quantity[i];
            r.timestamp = decltype(r.timestamp)(std::chrono::nanoseconds(now_ns[i]));
            reqs.push_back(std::move(r));
        }
        auto replies = handle.throttle_many(std::move(reqs));
        for (const Cmd& c : cmds_) {
            if (c.throttle < 0) {
                out += c.reply;
                continue;
            }
            const auto& r = replies[(size_t)c.throttle];
            if (r.index() == 0) { // [allowed, limit, remaining, reset_after s, retry_after s] (mod.rs:274-283)
                const auto& ok = std::get<0>(r);
                out += "*5\r\n";
                put_integer(out, ok.allowed ? 1 : 0);
                put_integer(out, ok.limit);
                put_integer(out, ok.remaining);
                put_integer(out, ok.reset_after);
                put_integer(out, ok.retry_after);
            } else {
                put_error(out, "ERR " + std::get<1>(r)); // mod.rs:285 `ERR {e}`
            }
        }
    }

    // replies of the commands that never reach the engine (for parser-only use)
    const std::string& immediate_reply(size_t cmd) const { return cmds_[cmd].reply; }
    bool is_throttle(size_t cmd) const { return cmds_[cmd].throttle >= 0; }

  private:
    struct Cmd {
        int64_t throttle = -1; // index into the request columns, or -1: `reply` is final
        std::string reply;
    };
    std::vector<Cmd> cmds_;

    void immediate(std::string r) {
        Cmd c;
        c.reply = std::move(r);
        cmds_.push_back(std::move(c));
    }
    void immediate_error(const std::string& msg) {
        std::string r;
        put_error(r, msg);
        immediate(std::move(r));
    }
    static bool arg_i64(const Elem& e, int64_t* v) { // mod.rs:289-295
        if (e.kind == Kind::Integer) {
            *v = e.integer;
            return true;
        }
        if (e.kind == Kind::Bulk) return parse_i64(std::string_view((const char*)e.p, e.n), v);
        return false;
    }

    // mod.rs:150-287
    void command(const Elem& top, const std::vector<Elem>& a, const Clock& clock) {
        if (top.kind != Kind::Array) return immediate_error("ERR expected array of commands");
        if (a.empty()) return immediate_error("ERR empty command");
        if (a[0].kind != Kind::Bulk) return immediate_error("ERR invalid command format");
        std::string name((const char*)a[0].p, a[0].n);
        for (char& ch : name)
            if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A'); // ASCII part of str::to_uppercase
        if (name == "PING") {
            std::string r;
            if (a.size() == 1) put_simple(r, "PONG");
            else if (a.size() == 2) put_elem(r, a[1]);
            else put_error(r, "ERR wrong number of arguments for 'ping' command");
            return immediate(std::move(r));
        }
        if (name == "QUIT") {
            std::string r;
            put_simple(r, "OK");
            quit = true;
            return immediate(std::move(r));
        }
        if (name != "THROTTLE") return immediate_error("ERR unknown command '" + name + "'");
        if (a.size() < 5 || a.size() > 6) return immediate_error("ERR wrong number of arguments for 'throttle' command");
        if (a[1].kind != Kind::Bulk) return immediate_error("ERR invalid key");
        int64_t burst, count, per, qty = 1;
        if (!arg_i64(a[2], &burst)) return immediate_error("ERR invalid max_burst");
        if (!arg_i64(a[3], &count)) return immediate_error("ERR invalid count_per_period");
        if (!arg_i64(a[4], &per)) return immediate_error("ERR invalid period");
        if (a.size() == 6 && !arg_i64(a[5], &qty)) return immediate_error("ERR invalid quantity");
        Cmd c;
        c.throttle = (int64_t)max_burst.size();
        cmds_.push_back(std::move(c));
        key_bytes.insert(key_bytes.end(), a[1].p, a[1].p + a[1].n);
        key_off.push_back((uint32_t)key_bytes.size());
        max_burst.push_back(burst);
        count_per_period.push_back(count);
        period.push_back(per);
        quantity.push_back(qty);
        now_ns.push_back(clock());
    }
};

} // namespace resp
} // namespace server
} // namespace throttlecrab

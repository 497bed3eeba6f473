// vfsx_host.h -- the host-only part of the library: error plumbing behind vfsx_last_error(), the walker / decoder
// of the reference's on-disk log (serializer.rs:25-153) and the host hit merge.  No HIP header is included here
// or in vfsx_host.cpp: that translation unit is built by plain g++ (and, for `make san`, with
// -fsanitize=address,undefined, which the device toolchain cannot run) because it parses an UNTRUSTED file.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "vfsx.h"

namespace vfsx {

// thread-local message behind vfsx_last_error()
void set_last_error(const char *msg);
inline void set_last_error(const std::string &msg) { set_last_error(msg.c_str()); }

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const std::string &msg) { throw Error(code, msg); }

// runs f; maps what it throws to a vfsx_status + the thread's last-error message.  Nothing unwinds across the ABI.
template <typename F>
int guarded(F &&f)
{
    try {
        set_last_error("");
        f();
        return VFSX_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return VFSX_ERR_INTERNAL;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return VFSX_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown failure");
        return VFSX_ERR_INTERNAL;
    }
}

// monotone u32 key of an f32 distance: the order of the reference's partial_cmp (rank.rs:170) with NaN last
inline uint32_t host_dist_key(float d)
{
    uint32_t b;
    __builtin_memcpy(&b, &d, 4);
    if ((b & 0x7fffffffu) > 0x7f800000u)
        return 0xffffffffu; // NaN sorts last
    if (b == 0x80000000u)
        b = 0; // -0.0 == +0.0 (partial_cmp)
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// VectorMetadata (vector.rs:11-17), kept host-side per row
struct RowMeta {
    std::string name;
    std::vector<std::string> tags;
    std::string created_at;
};

// one decoded record of the log; `values` points INTO the mapped file (unaligned)
struct LogRecord {
    uint32_t variant = 0; // 0 Dense, 1 Quantized (vector.rs:39-43)
    uint64_t id = 0;
    uint64_t n = 0;
    const uint8_t *values = nullptr; // n f32 (Dense) or n i8 (Quantized)
    float scale = 0.f;
    RowMeta meta;
};

// read-only mapping of a whole file; an empty file maps to (nullptr, 0)
struct MappedFile {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    explicit MappedFile(const char *path);
    ~MappedFile();
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

// bincode 1.3 (fixint, little endian) payload of one VFSVector -> r; throws Error(VFSX_ERR_SERIALIZATION)
void decode_record(const uint8_t *payload, size_t len, LogRecord &r);

// Walks `n` bytes the way load_vectors does (serializer.rs:101-150), minus its 1 KiB window: a marker starts a
// record, any other byte is skipped, an incomplete trailing record ends the walk.  Records with index < first are
// stepped over by their length only (not decoded); at most max_count records are delivered.  Returns the number
// delivered.
uint64_t walk_log(const uint8_t *p, size_t n, const std::function<void(const LogRecord &)> &on_record,
                  uint64_t first = 0, uint64_t max_count = ~0ull);

// vfsx_log_inspect over bytes already in memory (the entry point maps the file and calls this)
void inspect_log_bytes(const uint8_t *p, size_t n, vfsx_log_info *info);

void json_escape(std::string &out, const std::string &s);

} // namespace vfsx

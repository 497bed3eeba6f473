// vfsx_host.cpp -- host-only code of libvfsx.so, built by g++ (no HIP anywhere in this file): the thread's
// last-error message, the walker / decoder of the reference's append-only log and the host hit merge.
// `make san` builds this file alone with -fsanitize=address,undefined together with tests/cxx/log_fuzz.cpp
// (truncated, oversized, garbage and byte-mutated logs; tests/test_log_fuzz_cpu.py).
//
// F1: ingest from the reference's on-disk log (serializer.rs:25-153)
// record = DE AD BE EF | u64-LE payload length | bincode 1.3 (fixint, little endian) of VFSVector:
//   u32 variant (0 Dense, 1 Quantized) | u64 id | u64 n | n x f32 or n x i8 | [f32 scale_factor] |
//   String name | Vec<String> tags | String created_at (RFC 3339)      (String = u64 length + bytes)
#include "vfsx_host.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

thread_local std::string g_last_error;

// bincode decodes a String with String::from_utf8: bytes that are not UTF-8 fail the record.  Same rule here
// (overlong forms, surrogates and code points beyond U+10FFFF rejected), so that names and tags can go into
// JSON as they are.
bool valid_utf8(const uint8_t *s, size_t n)
{
    size_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) {
            ++i;
            continue;
        }
        size_t need;
        uint32_t cp, min;
        if ((c & 0xE0) == 0xC0) {
            need = 1, cp = c & 0x1F, min = 0x80;
        } else if ((c & 0xF0) == 0xE0) {
            need = 2, cp = c & 0x0F, min = 0x800;
        } else if ((c & 0xF8) == 0xF0) {
            need = 3, cp = c & 0x07, min = 0x10000;
        } else {
            return false;
        }
        if (n - i <= need)
            return false;
        for (size_t j = 1; j <= need; ++j) {
            if ((s[i + j] & 0xC0) != 0x80)
                return false;
            cp = (cp << 6) | (s[i + j] & 0x3F);
        }
        if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF))
            return false;
        i += need + 1;
    }
    return true;
}

// bounds-checked reader over one record's payload
struct Cursor {
    const uint8_t *p;
    size_t left;
    template <typename T>
    T get()
    {
        if (left < sizeof(T))
            vfsx::fail(VFSX_ERR_SERIALIZATION, "record payload truncated");
        T v;
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        left -= sizeof(T);
        return v;
    }
    const uint8_t *bytes(uint64_t n)
    {
        if (left < n)
            vfsx::fail(VFSX_ERR_SERIALIZATION, "record payload truncated");
        const uint8_t *r = p;
        p += n;
        left -= (size_t)n;
        return r;
    }
    std::string str()
    {
        const uint64_t n = get<uint64_t>();
        const uint8_t *b = bytes(n);
        if (!valid_utf8(b, (size_t)n))
            vfsx::fail(VFSX_ERR_SERIALIZATION, "record holds a string that is not UTF-8");
        return std::string((const char *)b, (size_t)n);
    }
};

} // namespace

namespace vfsx {

void set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

MappedFile::MappedFile(const char *path)
{
    fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        fail(VFSX_ERR_IO, std::string("cannot open ") + path + ": " + std::strerror(errno));
    // a constructor that throws runs no destructor: close the descriptor on every way out below
    struct stat st;
    if (fstat(fd, &st) != 0) {
        const int e = errno;
        ::close(fd);
        fd = -1;
        fail(VFSX_ERR_IO, std::string("cannot stat ") + path + ": " + std::strerror(e));
    }
    if (!S_ISREG(st.st_mode)) {
        ::close(fd);
        fd = -1;
        fail(VFSX_ERR_IO, std::string(path) + " is not a regular file");
    }
    n = (size_t)st.st_size;
    if (n) {
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            const int e = errno;
            ::close(fd);
            fd = -1;
            n = 0;
            fail(VFSX_ERR_IO, std::string("cannot map ") + path + ": " + std::strerror(e));
        }
        p = (const uint8_t *)m;
    }
}

MappedFile::~MappedFile()
{
    if (p)
        munmap((void *)p, n);
    if (fd >= 0)
        ::close(fd);
}

void decode_record(const uint8_t *payload, size_t len, LogRecord &r)
{
    Cursor c{payload, len};
    r.variant = c.get<uint32_t>();
    if (r.variant > 1)
        fail(VFSX_ERR_SERIALIZATION, "unknown VFSVector variant " + std::to_string(r.variant));
    r.id = c.get<uint64_t>();
    r.n = c.get<uint64_t>();
    if (r.n > len) // (also keeps n * 4 below from wrapping)
        fail(VFSX_ERR_SERIALIZATION, "vector length exceeds the record");
    r.values = c.bytes(r.n * (r.variant == 0 ? 4u : 1u));
    r.scale = r.variant == 1 ? c.get<float>() : 0.f;
    r.meta.name = c.str();
    const uint64_t nt = c.get<uint64_t>();
    if (nt > len / 8u) // every tag costs at least its 8-byte length
        fail(VFSX_ERR_SERIALIZATION, "tag count exceeds the record");
    r.meta.tags.clear();
    for (uint64_t i = 0; i < nt; ++i)
        r.meta.tags.push_back(c.str());
    r.meta.created_at = c.str();
}

uint64_t walk_log(const uint8_t *p, size_t n, const std::function<void(const LogRecord &)> &on_record, uint64_t first,
                  uint64_t max_count)
{
    static const uint8_t marker[4] = {0xDE, 0xAD, 0xBE, 0xEF}; // serializer.rs:13
    size_t cur = 0;
    uint64_t index = 0, count = 0;
    LogRecord r;
    while (n >= 4 && cur <= n - 4 && count < max_count) {
        if (std::memcmp(p + cur, marker, 4) != 0) {
            ++cur; // serializer.rs:148
            continue;
        }
        if (n - cur < 12)
            break; // serializer.rs:142-146
        uint64_t len;
        std::memcpy(&len, p + cur + 4, 8);
        if (len > n - (cur + 12))
            break; // serializer.rs:137-141
        if (index >= first) {
            decode_record(p + cur + 12, (size_t)len, r);
            on_record(r);
            ++count;
        }
        ++index;
        cur += 12 + (size_t)len;
    }
    return count;
}

void inspect_log_bytes(const uint8_t *p, size_t n, vfsx_log_info *info)
{
    vfsx_log_info out{};
    bool first = true;
    out.records = walk_log(p, n, [&](const LogRecord &r) {
        if (first) {
            out.dim = (uint32_t)r.n;
            out.dtype = r.variant == 1 ? VFSX_I8 : VFSX_F32;
            out.scale = r.scale;
            out.first_id = r.id;
            first = false;
        }
        out.last_id = r.id;
    });
    out.file_bytes = n;
    *info = out;
}

void json_escape(std::string &out, const std::string &s)
{
    out += '"';
    for (unsigned char ch : s) {
        switch (ch) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
            if (ch < 0x20) {
                char b[8];
                std::snprintf(b, sizeof b, "\\u%04x", ch);
                out += b;
            } else {
                out += (char)ch;
            }
        }
    }
    out += '"';
}

} // namespace vfsx

extern "C" {

const char *vfsx_last_error(void) { return g_last_error.c_str(); }

int vfsx_log_inspect(const char *path, vfsx_log_info *info)
{
    return vfsx::guarded([&] {
        if (!path || !info)
            vfsx::fail(VFSX_ERR_INVALID_VECTOR, "null argument");
        vfsx::MappedFile f(path);
        vfsx::inspect_log_bytes(f.p, f.n, info);
    });
}

int vfsx_quantize_rows_host(const float *values, uint64_t n, float scale, int8_t *out)
{
    return vfsx::guarded([&] {
        if (n && (!values || !out))
            vfsx::fail(VFSX_ERR_INVALID_VECTOR, "null buffer");
        for (uint64_t i = 0; i < n; ++i) {
            // Vector::quantize(Some(scale)) (vector.rs:94-108): (v * scale).round() as i8 -- one f32 multiply, round
            // half away from zero, Rust's saturating float -> int cast (NaN -> 0)
            const float r = std::round(values[i] * scale);
            out[i] = r != r ? (int8_t)0 : (r >= 127.0f ? (int8_t)127 : (r <= -128.0f ? (int8_t)-128 : (int8_t)r));
        }
    });
}

int vfsx_merge_hits_host(const vfsx_hit *in, uint32_t n_lists, uint32_t nq, uint32_t k, vfsx_hit *out)
{
    return vfsx::guarded([&] {
        if (!in || !out || n_lists == 0 || k == 0)
            vfsx::fail(VFSX_ERR_INVALID_VECTOR, "bad merge arguments");
        std::vector<vfsx_hit> tmp((size_t)n_lists * k);
        for (uint32_t q = 0; q < nq; ++q) {
            for (uint32_t l = 0; l < n_lists; ++l)
                std::memcpy(&tmp[(size_t)l * k], &in[((size_t)l * nq + q) * k], k * sizeof(vfsx_hit));
            std::stable_sort(tmp.begin(), tmp.end(), [](const vfsx_hit &a, const vfsx_hit &b) {
                const uint32_t va = a.flags & 1u, vb = b.flags & 1u;
                if (va != vb)
                    return va > vb;
                const uint32_t ka = vfsx::host_dist_key(a.dist), kb = vfsx::host_dist_key(b.dist);
                if (ka != kb)
                    return ka < kb;
                return a.row < b.row;
            });
            std::memcpy(&out[(size_t)q * k], tmp.data(), k * sizeof(vfsx_hit));
        }
    });
}

} // extern "C"

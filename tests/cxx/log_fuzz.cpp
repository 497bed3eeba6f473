// log_fuzz.cpp -- CPU fuzz driver for the host-only log walker / decoder (vfs_amd/csrc/vfsx_host.cpp), built
// together with it under -fsanitize=address,undefined by `make -C vfs_amd/csrc san`.  Test infrastructure.
//
//   log_fuzz_san <golden log> <scratch dir>
//
// Feeds the parser the reference's own first records (tests/golden/vectors_head.dat) and hostile variants of
// them: every truncation, every byte mutated, oversized length / element-count / tag-count / string-length
// fields, garbage with and without markers, an empty file, a directory, a missing file.  Every call must return
// VFSX_OK (possibly with a short record count: serializer.rs:137-146 ends the walk at an incomplete record) or
// VFSX_ERR_SERIALIZATION / VFSX_ERR_IO with a message -- and the sanitizers must stay silent (they abort the
// process otherwise: -fno-sanitize-recover=all).  Prints one summary line; exit status 0 = all expectations held.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <sys/resource.h>
#include <unistd.h>

#include "vfsx_host.h"

namespace {

struct Tally {
    uint64_t ok = 0, short_ok = 0, ser = 0, io = 0, other = 0;
};

int failures = 0;

void expect(bool cond, const char *what)
{
    if (!cond) {
        std::fprintf(stderr, "log_fuzz: expectation failed: %s\n", what);
        ++failures;
    }
}

// parse bytes in memory, through the same guarded() mapping the entry points use
int parse(const std::vector<uint8_t> &b, vfsx_log_info *info, Tally &t, uint64_t full_records)
{
    // hand the parser an exact-size heap copy so that ASan sees any read past the end
    uint8_t *copy = b.empty() ? nullptr : (uint8_t *)std::malloc(b.size());
    if (copy)
        std::memcpy(copy, b.data(), b.size());
    const int rc = vfsx::guarded([&] { vfsx::inspect_log_bytes(copy, b.size(), info); });
    std::free(copy);
    if (rc == VFSX_OK)
        (info->records == full_records ? t.ok : t.short_ok) += 1;
    else if (rc == VFSX_ERR_SERIALIZATION)
        t.ser += 1;
    else if (rc == VFSX_ERR_IO)
        t.io += 1;
    else
        t.other += 1;
    if (rc != VFSX_OK)
        expect(vfsx_last_error()[0] != 0, "an error carries a message");
    return rc;
}

void put_u64(std::vector<uint8_t> &b, size_t off, uint64_t v) { std::memcpy(b.data() + off, &v, 8); }

uint64_t get_u64(const std::vector<uint8_t> &b, size_t off)
{
    uint64_t v;
    std::memcpy(&v, b.data() + off, 8);
    return v;
}

void write_file(const std::string &path, const std::vector<uint8_t> &b)
{
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    f.write((const char *)b.data(), (std::streamsize)b.size());
}

int open_fds()
{
    int n = 0;
    for (int fd = 0; fd < 256; ++fd)
        if (::lseek(fd, 0, SEEK_CUR) != -1 || errno != EBADF)
            ++n;
    return n;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <golden log> <scratch dir>\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[1], std::ios::binary);
    std::vector<uint8_t> gold((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const std::string dir = argv[2];
    expect(gold.size() > 100, "golden log read");

    Tally t;
    vfsx_log_info info{};
    // 0. the untouched file
    expect(parse(gold, &info, t, ~0ull) == VFSX_OK && info.records >= 2, "golden log parses");
    const uint64_t full = info.records;
    const uint32_t dim = info.dim;
    t = Tally();

    // 1. every truncation (mid-marker, mid-length, mid-payload, mid-string): a clean short count, never an error
    for (size_t n = 0; n <= gold.size(); ++n) {
        std::vector<uint8_t> b(gold.begin(), gold.begin() + (long)n);
        const int rc = parse(b, &info, t, full);
        expect(rc == VFSX_OK, "a truncated log ends the walk cleanly (serializer.rs:137-146)");
        expect(info.records <= full, "a truncated log never yields more records");
    }

    // 2. every byte, eight hostile values
    for (size_t i = 0; i < gold.size(); ++i) {
        const uint8_t o = gold[i];
        const uint8_t vals[8] = {0x00, 0xFF, 0x7F, 0x80, (uint8_t)(o ^ 1), (uint8_t)(o ^ 0x80), (uint8_t)(o + 1), 0xDE};
        for (uint8_t v : vals) {
            if (v == o)
                continue;
            std::vector<uint8_t> b = gold;
            b[i] = v;
            const int rc = parse(b, &info, t, full);
            expect(rc == VFSX_OK || rc == VFSX_ERR_SERIALIZATION, "a mutated byte gives OK or a serialization error");
        }
    }

    // 3. targeted fields of the first record: marker(4) len(8) | variant(4) id(8) n(8) values..
    const uint64_t huge[] = {~0ull, 1ull << 63, 1ull << 32, (1ull << 32) - 1, gold.size(), gold.size() - 11, 0x7fffffffffffffffull};
    for (uint64_t h : huge) {
        std::vector<uint8_t> b = gold;
        put_u64(b, 4, h); // record length
        expect(parse(b, &info, t, full) == VFSX_OK, "an oversized record length ends the walk (serializer.rs:137-141)");
        b = gold;
        put_u64(b, 12 + 4 + 8, h); // element count n
        const int rc = parse(b, &info, t, full);
        expect(rc == VFSX_ERR_SERIALIZATION, "an oversized element count is a serialization error");
        // string length of `name` and the tag count sit after the values (+ scale for Quantized)
        const uint32_t variant = (uint32_t)gold[12];
        const size_t name_off = 12 + 4 + 8 + 8 + (size_t)dim * (variant ? 1 : 4) + (variant ? 4 : 0);
        b = gold;
        put_u64(b, name_off, h);
        expect(parse(b, &info, t, full) == VFSX_ERR_SERIALIZATION, "an oversized string length is a serialization error");
        const size_t tags_off = name_off + 8 + (size_t)get_u64(gold, name_off);
        b = gold;
        put_u64(b, tags_off, h);
        expect(parse(b, &info, t, full) == VFSX_ERR_SERIALIZATION, "an oversized tag count is a serialization error");
    }
    {
        std::vector<uint8_t> b = gold;
        b[12] = 2; // unknown enum variant
        expect(parse(b, &info, t, full) == VFSX_ERR_SERIALIZATION, "an unknown variant is a serialization error");
        b = gold;
        const uint32_t variant = (uint32_t)gold[12];
        const size_t name_off = 12 + 4 + 8 + 8 + (size_t)dim * (variant ? 1 : 4) + (variant ? 4 : 0);
        b[name_off + 8] = 0xC0; // invalid UTF-8 in the name (bincode: String::from_utf8 fails)
        expect(parse(b, &info, t, full) == VFSX_ERR_SERIALIZATION, "a non-UTF-8 name is a serialization error");
    }

    // 4. garbage: no marker at all, markers everywhere, a marker as the last bytes, random bytes
    {
        std::vector<uint8_t> b(4096, 0xAB);
        expect(parse(b, &info, t, full) == VFSX_OK && info.records == 0, "garbage without a marker holds no records");
        for (size_t i = 0; i + 4 <= b.size(); i += 4) {
            b[i] = 0xDE, b[i + 1] = 0xAD, b[i + 2] = 0xBE, b[i + 3] = 0xEF;
        }
        const int rc = parse(b, &info, t, full);
        expect(rc == VFSX_OK || rc == VFSX_ERR_SERIALIZATION, "a file of markers is survivable");
        std::vector<uint8_t> tail = gold;
        const uint8_t m[4] = {0xDE, 0xAD, 0xBE, 0xEF};
        tail.insert(tail.end(), m, m + 4);
        expect(parse(tail, &info, t, full) == VFSX_OK && info.records == full, "a trailing bare marker ends the walk");
        for (int k = 1; k < 12; ++k) { // marker + part of a length
            std::vector<uint8_t> b2 = tail;
            b2.insert(b2.end(), (size_t)k, 0x01);
            expect(parse(b2, &info, t, full) == VFSX_OK, "a trailing partial header ends the walk");
        }
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int round = 0; round < 200; ++round) {
            std::vector<uint8_t> r((size_t)(round * 37 % 3000));
            for (auto &x : r) {
                s ^= s << 13, s ^= s >> 7, s ^= s << 17;
                x = (uint8_t)s;
            }
            // seed a few markers + plausible small lengths so that the decoder is actually reached
            for (size_t i = 0; i + 16 < r.size(); i += 97) {
                std::memcpy(&r[i], m, 4);
                put_u64(r, i + 4, (uint64_t)(r[i + 12] % 80));
            }
            const int rc2 = parse(r, &info, t, full);
            expect(rc2 == VFSX_OK || rc2 == VFSX_ERR_SERIALIZATION, "random bytes give OK or a serialization error");
        }
    }

    // 5. through the file entry point: empty file, truncated file, directory, missing file -- and no descriptor leaks
    const int fds_before = open_fds();
    {
        write_file(dir + "/empty.dat", {});
        expect(vfsx_log_inspect((dir + "/empty.dat").c_str(), &info) == VFSX_OK && info.records == 0 && info.file_bytes == 0,
               "a 0-byte file holds no records");
        std::vector<uint8_t> half(gold.begin(), gold.begin() + (long)(gold.size() / 2));
        write_file(dir + "/half.dat", half);
        expect(vfsx_log_inspect((dir + "/half.dat").c_str(), &info) == VFSX_OK && info.records < full,
               "a file cut mid-record yields the whole records before the cut");
        write_file(dir + "/gold.dat", gold);
        expect(vfsx_log_inspect((dir + "/gold.dat").c_str(), &info) == VFSX_OK && info.records == full, "file == memory");
        for (int i = 0; i < 64; ++i) { // failing opens / maps must not leak descriptors
            expect(vfsx_log_inspect(dir.c_str(), &info) == VFSX_ERR_IO, "a directory is an I/O error");
            expect(vfsx_log_inspect((dir + "/missing.dat").c_str(), &info) == VFSX_ERR_IO, "a missing file is an I/O error");
        }
        expect(vfsx_log_inspect(nullptr, &info) == VFSX_ERR_INVALID_VECTOR, "null path");
        expect(vfsx_log_inspect((dir + "/gold.dat").c_str(), nullptr) == VFSX_ERR_INVALID_VECTOR, "null info");
    }
    expect(open_fds() == fds_before, "no file descriptor leaked");

    // 6. the host merge on degenerate input
    {
        vfsx_hit in[4] = {}, out[2] = {};
        expect(vfsx_merge_hits_host(nullptr, 1, 1, 1, out) == VFSX_ERR_INVALID_VECTOR, "merge: null input");
        expect(vfsx_merge_hits_host(in, 0, 1, 1, out) == VFSX_ERR_INVALID_VECTOR, "merge: no lists");
        expect(vfsx_merge_hits_host(in, 2, 0, 2, out) == VFSX_OK, "merge: no queries");
        in[0].dist = 2.f, in[0].flags = 1, in[0].row = 5;
        in[2].dist = 2.f, in[2].flags = 1, in[2].row = 3;
        expect(vfsx_merge_hits_host(in, 2, 1, 2, out) == VFSX_OK && out[0].row == 3 && out[1].row == 5, "merge: tie by row");
    }

    std::printf("{\"cases_ok\": %llu, \"cases_short\": %llu, \"cases_serialization_error\": %llu, \"cases_io_error\": %llu, "
                "\"cases_other\": %llu, \"failed_expectations\": %d, \"golden_records\": %llu}\n",
                (unsigned long long)t.ok, (unsigned long long)t.short_ok, (unsigned long long)t.ser,
                (unsigned long long)t.io, (unsigned long long)t.other, failures, (unsigned long long)full);
    return failures == 0 && t.other == 0 ? 0 : 1;
}

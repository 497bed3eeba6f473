// vfsx_ranker.hpp -- C++ host-side mirror of the reference's Ranker interface
// (vfs/src/vfs/rank.rs:61-105) above the C ABI of include/vfsx.h.  Header-only; same
// names, argument meaning and error behaviour as the Rust original, with the
// device-resident vfsx::Corpus standing where VFSManager did on this path.
#pragma once

#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vfsx.h"

namespace vfsx {

// VFSError (vfs/src/vfs/err.rs:10-16) as an exception carrying the vfsx_status.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc)
{
    if (rc != VFSX_OK)
        throw Error(rc, vfsx_last_error());
}

enum class SearchType { Exact, Approximate };                                   // rank.rs:61-64
enum class DistanceMethod { Euclidean, Cosine, SimdEuclidean, SimdCosine,     // rank.rs:67-72
                            EuclideanI8Domain /* extension: VFSX_EUCLIDEAN_I8DOMAIN, int8 corpora only */ };

// The vector set the scan reads, in scan order (what load_batch yields,
// storage_manager.rs:166-217), resident in HBM.
class Corpus {
public:
    Corpus(uint32_t dim, bool quantize = false, float scale = 127.0f, uint64_t capacity_rows = 0, int device = 0)
        : dim_(dim), quantize_(quantize)
    {
        h_ = vfsx_corpus_create(dim, quantize ? VFSX_I8 : VFSX_F32, scale, capacity_rows, device);
        if (!h_)
            throw Error(VFSX_ERR_NO_DEVICE, vfsx_last_error());
    }
    ~Corpus() { vfsx_corpus_destroy(h_); }
    Corpus(const Corpus &) = delete;
    Corpus &operator=(const Corpus &) = delete;

    uint32_t dim() const { return dim_; }
    bool quantized() const { return quantize_; }
    uint64_t rows() const { return vfsx_corpus_rows(h_); }
    vfsx_corpus *handle() const { return h_; }

    // register_vector_from_vec (storage_manager.rs:317-335): f32 in; a quantizing store
    // quantizes on flush (storage_manager.rs:143-147) -- here on the device.
    void register_vectors(const float *rows, uint64_t n, const uint64_t *ids = nullptr)
    {
        check(quantize_ ? vfsx_corpus_append_quantize(h_, rows, ids, n) : vfsx_corpus_append(h_, rows, ids, n));
    }
    // already-stored QuantizedVector bytes
    void append_quantized(const int8_t *rows, uint64_t n, const uint64_t *ids = nullptr)
    {
        if (!quantize_)
            throw Error(VFSX_ERR_INVALID_VECTOR, "corpus is not quantized");
        check(vfsx_corpus_append(h_, rows, ids, n));
    }

    // serializer.rs:77-153 / storage_manager.rs:166-217: stream an existing data/vectors.dat into HBM
    // (ids and metadata as recorded); returns the records loaded
    uint64_t load_log(const std::string &path)
    {
        uint64_t n = 0;
        check(vfsx_corpus_load_log(h_, path.c_str(), &n));
        return n;
    }
    // this shard's first global scan position (row-sharded stores)
    void set_row_base(uint64_t row_base) { check(vfsx_corpus_set_row_base(h_, row_base)); }
    // many-query batches on the matrix cores (no reference counterpart; results unchanged)
    // kind: 0 off, 1 bf16 mirror, 2 int8 mirror of an f32 store, 3 the int8 store itself
    std::pair<int, float> mfma_mirror()
    {
        int kind = 0;
        float res = 0.0f;
        check(vfsx_corpus_mfma_mirror(h_, &kind, &res));
        return {kind, res};
    }
    // mirror (f32 stores): 1 auto, 2 bf16, 3 int8 with a scale per row (include/vfsx.h)
    void enable_mfma(bool on = true, int mirror = 1) { check(vfsx_corpus_enable_mfma(h_, on ? mirror : 0)); }

    // nq queries at once ("A2/A3 + A4 per query", SURVEY.md 8a batch note): flat [nq][k] outputs
    struct Batch {
        std::vector<uint64_t> ids;
        std::vector<float> dist;
        std::vector<uint32_t> count;
    };
    Batch search_batch(const float *queries, uint32_t nq, uint32_t k, int metric)
    {
        Batch b;
        b.ids.resize((size_t)nq * k);
        b.dist.resize((size_t)nq * k);
        b.count.resize(nq);
        check(vfsx_search(h_, queries, nq, k, metric, b.ids.data(), b.dist.data(), b.count.data()));
        return b;
    }

private:
    vfsx_corpus *h_ = nullptr;
    uint32_t dim_;
    bool quantize_;
};

// One rank's handle on the RCCL communicator of a row-sharded store (include/vfsx.h, vfsx_comm_*; no
// reference counterpart).  Rank 0 makes the id, the host hands it to every rank, all ranks construct.
class Comm {
public:
    using Id = std::array<char, VFSX_COMM_ID_BYTES>;
    static Id unique_id()
    {
        Id id{};
        check(vfsx_comm_unique_id(id.data()));
        return id;
    }
    Comm(int rank, int world, const Id &id, int device = 0)
    {
        h_ = vfsx_comm_init(rank, world, id.data(), device);
        if (!h_)
            throw Error(VFSX_ERR_INTERNAL, vfsx_last_error());
    }
    ~Comm() { vfsx_comm_destroy(h_); }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    int rank() const { return vfsx_comm_rank(h_); }
    int world() const { return vfsx_comm_world(h_); }
    vfsx_comm *handle() const { return h_; }

private:
    vfsx_comm *h_ = nullptr;
};

class Ranker {
public:
    // Ranker::new (rank.rs:85-87)
    Ranker(SearchType search_type, DistanceMethod distance_method, std::optional<size_t> ef_search = std::nullopt)
        : search_type_(search_type), distance_method_(distance_method), ef_search_(ef_search.value_or(6))
    {
    }

    // Ranker::search (rank.rs:90-105): (id, distance) pairs, ascending, <= result_limit.
    std::vector<std::pair<uint64_t, float>> search(const std::vector<float> &query, size_t num_vectors_per_iteration,
                                                   std::optional<size_t> result_limit, Corpus &manager) const
    {
        if (search_type_ != SearchType::Exact)
            throw Error(VFSX_ERR_INTERNAL, "approximate (HNSW) search is outside the vfsx hot path");
        if (query.size() != manager.dim()) // rank.rs:273-275 panics
            throw Error(VFSX_ERR_INVALID_VECTOR, "Los vectores deben tener la misma dimensión");
        const size_t limit = result_limit.value_or(5); // rank.rs:129
        if (num_vectors_per_iteration == 0)            // load_batch(0) yields nothing
            return {};
        if (limit == 0) // rank.rs:175: last().unwrap() on an empty Vec panics
            throw Error(VFSX_ERR_INVALID_VECTOR, "result_limit must be >= 1");
        std::vector<uint64_t> ids(limit);
        std::vector<float> dist(limit);
        uint32_t count = 0;
        check(vfsx_search(manager.handle(), query.data(), 1, (uint32_t)limit, (int)distance_method_, ids.data(),
                          dist.data(), &count));
        std::vector<std::pair<uint64_t, float>> out;
        out.reserve(count);
        for (uint32_t i = 0; i < count; ++i)
            out.emplace_back(ids[i], dist[i]);
        return out;
    }

    // The same search over a row-sharded store: `shard` is this rank's contiguous range of the scan order
    // (Corpus::set_row_base), every rank calls with the same query and gets the same global ranking
    // (local scan -> ncclAllGather -> merge by (distance, global row)).
    std::vector<std::pair<uint64_t, float>> search_sharded(const std::vector<float> &query, size_t num_vectors_per_iteration,
                                                           std::optional<size_t> result_limit, Corpus &shard,
                                                           Comm &comm) const
    {
        if (search_type_ != SearchType::Exact)
            throw Error(VFSX_ERR_INTERNAL, "approximate (HNSW) search is outside the vfsx hot path");
        if (query.size() != shard.dim())
            throw Error(VFSX_ERR_INVALID_VECTOR, "Los vectores deben tener la misma dimensión");
        const size_t limit = result_limit.value_or(5);
        if (limit == 0)
            throw Error(VFSX_ERR_INVALID_VECTOR, "result_limit must be >= 1");
        // num_vectors_per_iteration == 0 (empty result) must still join the collective: every rank calls
        std::vector<uint64_t> ids(limit);
        std::vector<float> dist(limit);
        uint32_t count = 0;
        check(vfsx_search_sharded_host(shard.handle(), comm.handle(), query.data(), 1, (uint32_t)limit,
                                       (int)distance_method_, ids.data(), dist.data(), &count));
        std::vector<std::pair<uint64_t, float>> out;
        if (num_vectors_per_iteration == 0)
            return out;
        out.reserve(count);
        for (uint32_t i = 0; i < count; ++i)
            out.emplace_back(ids[i], dist[i]);
        return out;
    }

    size_t ef_search() const { return ef_search_; }

private:
    SearchType search_type_;
    DistanceMethod distance_method_;
    size_t ef_search_;
};

} // namespace vfsx

// Program wire format: a stable byte image of gaast_program_desc (include/gaast_expr.h).
#include <cstring>
#include <memory>
#include <vector>

#include "../common/algebra.hpp"
#include "gaast_expr.h"

namespace {

constexpr char MAGIC[8] = {'G', 'A', 'A', 'S', 'T', 'P', 'R', 'G'};
constexpr uint32_t VERSION = 1;

struct Writer {
    unsigned char* buf;
    size_t cap, pos = 0;
    void put(const void* p, size_t n) {
        if (buf && pos + n <= cap) std::memcpy(buf + pos, p, n);
        pos += n;
    }
    template <class T>
    void val(T v) { put(&v, sizeof(T)); }
};

struct Reader {
    const unsigned char* buf;
    size_t len, pos = 0;
    bool ok = true;
    bool get(void* p, size_t n) {
        if (!ok || pos + n > len) return ok = false;
        std::memcpy(p, buf + pos, n);
        pos += n;
        return true;
    }
    template <class T>
    T val() {
        T v{};
        get(&v, sizeof(T));
        return v;
    }
};

}  // namespace

struct gaast_program_image_s {
    gaast_program_desc desc{};
    std::vector<double> metric;
    std::vector<gaast_node_desc> nodes;
    std::vector<std::vector<gaast_comp_mul>> lists;
    std::vector<gaast_input_desc> inputs;
    std::vector<std::vector<double>> const_rows;
};

extern "C" {

size_t gaast_program_serialize(const gaast_program_desc* d, void* buf, size_t cap) {
    Writer w{static_cast<unsigned char*>(buf), cap};
    w.put(MAGIC, 8);
    w.val<uint32_t>(VERSION);
    w.val<int32_t>(d->vec_space_dim);
    w.val<int32_t>(d->dtype);
    w.val<uint32_t>(d->flags);
    w.val<int32_t>(d->root);
    for (int i = 0; i < d->vec_space_dim; ++i) w.val<double>(d->metric_diag[i]);
    w.val<int32_t>(d->n_inputs);
    for (int i = 0; i < d->n_inputs; ++i) {
        const gaast_input_desc& in = d->inputs[i];
        w.val<uint64_t>(in.grade_mask);
        w.val<int32_t>(in.storage_dim);
        w.val<int32_t>(in.is_const);
        if (in.is_const) {
            const int64_t len = gaast::row_len_of(in.storage_dim, in.grade_mask);
            for (int64_t c = 0; c < len; ++c) w.val<double>(in.const_row[c]);
        }
    }
    w.val<int32_t>(d->n_nodes);
    for (int i = 0; i < d->n_nodes; ++i) {
        const gaast_node_desc& n = d->nodes[i];
        w.val<int32_t>(n.opcode);
        w.val<int32_t>(n.child0);
        w.val<int32_t>(n.child1);
        w.val<uint64_t>(n.minimal_grade_mask);
        w.val<int32_t>(n.vec_space_dim);
        w.val<int32_t>(n.input_slot);
        w.val<int32_t>(n.product_kind);
        w.val<uint64_t>(n.n_comp_muls);
        const uint8_t has_list = n.comp_muls != nullptr && n.n_comp_muls > 0;
        w.val<uint8_t>(has_list);
        if (has_list) w.put(n.comp_muls, size_t(n.n_comp_muls) * sizeof(gaast_comp_mul));
    }
    return w.pos;
}

gaast_program_image_t gaast_program_deserialize(const void* buf, size_t len) {
    Reader r{static_cast<const unsigned char*>(buf), len};
    char magic[8];
    if (!r.get(magic, 8) || std::memcmp(magic, MAGIC, 8) != 0) return nullptr;
    if (r.val<uint32_t>() != VERSION) return nullptr;
    auto img = std::make_unique<gaast_program_image_s>();
    gaast_program_desc& d = img->desc;
    d.vec_space_dim = r.val<int32_t>();
    d.dtype = r.val<int32_t>();
    d.flags = r.val<uint32_t>();
    d.root = r.val<int32_t>();
    if (!r.ok || d.vec_space_dim < 0 || d.vec_space_dim > GAAST_MAX_DIM) return nullptr;
    for (int i = 0; i < d.vec_space_dim; ++i) img->metric.push_back(r.val<double>());
    d.n_inputs = r.val<int32_t>();
    if (!r.ok || d.n_inputs < 0 || d.n_inputs > GAAST_MAX_INPUTS) return nullptr;
    img->const_rows.resize(size_t(d.n_inputs));
    for (int i = 0; i < d.n_inputs; ++i) {
        gaast_input_desc in{};
        in.grade_mask = r.val<uint64_t>();
        in.storage_dim = r.val<int32_t>();
        in.is_const = r.val<int32_t>();
        if (!r.ok || in.storage_dim < 0 || in.storage_dim > GAAST_MAX_DIM) return nullptr;
        if (in.is_const) {
            const int64_t n = gaast::row_len_of(in.storage_dim, in.grade_mask);
            for (int64_t c = 0; c < n; ++c) img->const_rows[size_t(i)].push_back(r.val<double>());
        }
        img->inputs.push_back(in);
    }
    d.n_nodes = r.val<int32_t>();
    if (!r.ok || d.n_nodes <= 0 || d.n_nodes > (1 << 20)) return nullptr;
    img->lists.resize(size_t(d.n_nodes));
    for (int i = 0; i < d.n_nodes; ++i) {
        gaast_node_desc n{};
        n.opcode = r.val<int32_t>();
        n.child0 = r.val<int32_t>();
        n.child1 = r.val<int32_t>();
        n.minimal_grade_mask = r.val<uint64_t>();
        n.vec_space_dim = r.val<int32_t>();
        n.input_slot = r.val<int32_t>();
        n.product_kind = r.val<int32_t>();
        n.n_comp_muls = r.val<uint64_t>();
        const uint8_t has_list = r.val<uint8_t>();
        if (!r.ok) return nullptr;
        if (has_list) {
            if (n.n_comp_muls > (len - r.pos) / sizeof(gaast_comp_mul)) return nullptr;
            img->lists[size_t(i)].resize(size_t(n.n_comp_muls));
            if (!r.get(img->lists[size_t(i)].data(), size_t(n.n_comp_muls) * sizeof(gaast_comp_mul))) return nullptr;
        }
        img->nodes.push_back(n);
    }
    if (!r.ok || r.pos != len) return nullptr;
    for (int i = 0; i < d.n_inputs; ++i)
        img->inputs[size_t(i)].const_row = img->inputs[size_t(i)].is_const ? img->const_rows[size_t(i)].data() : nullptr;
    for (int i = 0; i < d.n_nodes; ++i)
        img->nodes[size_t(i)].comp_muls = img->lists[size_t(i)].empty() ? nullptr : img->lists[size_t(i)].data();
    d.metric_diag = img->metric.data();
    d.inputs = img->inputs.data();
    d.nodes = img->nodes.data();
    return img.release();
}

const gaast_program_desc* gaast_program_image_desc(gaast_program_image_t img) { return img ? &img->desc : nullptr; }
void gaast_program_image_free(gaast_program_image_t img) { delete img; }

}  // extern "C"

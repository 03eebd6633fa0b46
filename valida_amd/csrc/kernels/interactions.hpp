// Device encoding of a chip's bus interactions (Chip::all_interactions, machine/src/chip.rs:40-63;
// VirtualPairCol affine forms) shared by the permutation-trace kernels and the quotient kernel.
//
// Flat u32 word stream (uniform/scalar loads on the device):
//   [0] M  (number of interactions)     [1] max_fields
//   then per interaction m:  [sign (1 = send, 0 = receive)] [n_fields] count_vcol fields_vcol...
//   vcol := [n_terms] [constant (Montgomery)] then n_terms x { [col | is_prep << 31] [weight (Montgomery)] }
// Interaction m's words start at offsets[m] (offset table follows the header: words [2 .. 2+M)).
#pragma once
#include <vector>
#include "../air/builder.hpp"
#include "../field.hpp"

namespace vk {

inline void encode_vcol(std::vector<uint32_t>& w, const vair::VirtualCol& v) {
    w.push_back((uint32_t)v.terms.size());
    w.push_back(vg::Fp::from_canonical(v.constant).v);
    for (auto& t : v.terms) {
        w.push_back((uint32_t)t.col | (t.preprocessed ? 0x80000000u : 0));
        w.push_back(vg::Fp::from_canonical(t.weight).v);
    }
}

inline std::vector<uint32_t> encode_interactions(const std::vector<vair::Interaction>& its) {
    std::vector<uint32_t> w;
    uint32_t M = (uint32_t)its.size(), maxf = 0;
    for (auto& it : its) maxf = std::max<uint32_t>(maxf, (uint32_t)it.fields.size());
    w.push_back(M);
    w.push_back(maxf);
    size_t table = w.size();
    w.resize(w.size() + M);
    for (uint32_t m = 0; m < M; m++) {
        w[table + m] = (uint32_t)w.size();
        w.push_back(its[m].is_send() ? 1u : 0u);
        w.push_back((uint32_t)its[m].fields.size());
        encode_vcol(w, its[m].count);
        for (auto& f : its[m].fields) encode_vcol(w, f);
    }
    return w;
}

#if defined(__HIPCC__)
}  // namespace vk
#include "device_common.hpp"  // load_at
namespace vk {
// Evaluate the vcol at `*pos` on one row; advances *pos past it.  `main`/`prep` are column-major views
// (data, stride) and `row` the storage row.
__device__ __forceinline__ vg::Fp eval_vcol(const uint32_t* __restrict__ w, uint32_t& pos, const uint32_t* main, uint64_t mstride,
                                            const uint32_t* prep, uint64_t pstride, uint64_t row) {
    uint32_t nt = w[pos];
    vg::Fp acc = vg::Fp::raw(w[pos + 1]);
    pos += 2;
    for (uint32_t t = 0; t < nt; t++, pos += 2) {
        uint32_t cw = w[pos];
        uint32_t col = cw & 0x7fffffffu;
        vg::Fp v = (cw >> 31) ? vg::Fp::raw(prep[(uint64_t)col * pstride + row]) : vg::Fp::raw(main[(uint64_t)col * mstride + row]);
        const uint32_t wt = w[pos + 1];  // wave-uniform: almost every field is a bare column (weight 1)
        acc += wt == vg::R_MOD_P ? v : v * vg::Fp::raw(wt);
    }
    return acc;
}
// The same with the row given as a 32-bit BYTE offset into every column: the column's base (data + col * stride) is wave-uniform, the load a raw buffer
// load (device_common.hpp: load_at).  LDE heights are at most 2^27 rows (the field's two-adicity), so the offset is below the resource's 2^31-byte range.
__device__ __forceinline__ vg::Fp eval_vcol_at(const uint32_t* __restrict__ w, uint32_t& pos, const uint32_t* main, uint64_t mstride,
                                               const uint32_t* prep, uint64_t pstride, uint32_t byte_off) {
    uint32_t nt = w[pos];
    vg::Fp acc = vg::Fp::raw(w[pos + 1]);
    pos += 2;
    for (uint32_t t = 0; t < nt; t++, pos += 2) {
        uint32_t cw = w[pos];
        uint32_t col = cw & 0x7fffffffu;
        vg::Fp v = vg::Fp::raw((cw >> 31) ? load_at(prep + (uint64_t)col * pstride, byte_off) : load_at(main + (uint64_t)col * mstride, byte_off));
        const uint32_t wt = w[pos + 1];
        acc += wt == vg::R_MOD_P ? v : v * vg::Fp::raw(wt);
    }
    return acc;
}
#endif

}  // namespace vk

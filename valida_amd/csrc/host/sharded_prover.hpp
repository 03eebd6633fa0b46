// ONE proof over W GPUs (intra-proof sharding): Machine::prove (basic/src/lib.rs:147-675) with every committed
// LDE, every Merkle tree, the quotient evaluation, the openings and the FRI commit phase held and computed in ROW-RANGE shards, one
// per rank, and the very proof words the single-GPU prover produces.  Design: sharded_prover.cpp.
#pragma once
#include "fabric.hpp"
#include "prover.hpp"

namespace vhost {

// The traces of one rank.  Two forms:
//   replicated (full_height empty)  every rank holds the whole main / preprocessed traces; the large objects of a proof — LDEs, trees,
//                                   reduced openings, FRI layers — are what is sharded;
//   row ranges (full_height[i] = the chip's whole trace height)  a chip whose LDE is sharded (at least max(4 W, 2^log_min_sharded, 2 blowup)
//                                   rows: sharded_trace_is_split) hands in ONLY the rows [rank n / W, (rank + 1) n / W) of its main trace —
//                                   main[i]->height = full_height[i] / W — and every other chip its whole trace (height = full_height[i]).
//                                   Per-rank trace memory and the permutation-trace work are then 1 / W as well: the running sum of
//                                   generate_permutation_trace (machine/src/chip.rs:176-205) is a local scan plus ONE exchange of the ranks'
//                                   totals, and the commitment rounds deal the row ranges into whole columns with one more all-to-all.
//                                   The preprocessed traces (program ROM, range table: constants of the program) stay whole on every rank.
struct ShardedInputs {
    std::vector<const DeviceTrace*> main;
    std::vector<std::pair<int, const DeviceTrace*>> prep;
    std::vector<uint64_t> full_height;
};
// whether a chip of this trace height is handed in as row ranges (the rule every rank and the library apply)
inline bool sharded_trace_is_split(uint32_t world, unsigned log_blowup, unsigned log_min_sharded, uint64_t height) {
    if (world < 2) return false;
    const uint64_t min_big = std::max<uint64_t>(std::max<uint64_t>(4ull * world, 1ull << log_min_sharded), 2ull << log_blowup);
    return (height << log_blowup) >= min_big;
}

struct ShardedProof {
    // provers[k] / in[k]: the prover context and traces of hosted rank fabric.hosted[k].  Matrices whose LDE has at least
    // max(4 W, 2^log_min_sharded) rows are sharded, shorter ones are computed by every rank.  Returns the proof words (the same on every rank).
    // Multi-process fabrics: if this rank throws, its peers are told (Fabric::fail) and throw FabricPeerFailure at their next collective.
    static std::vector<uint32_t> run(Fabric& fabric, const std::vector<Prover*>& provers, const std::vector<ShardedInputs>& in, unsigned log_min_sharded);

  private:
    static std::vector<uint32_t> run_impl(Fabric& fabric, const std::vector<Prover*>& provers, const std::vector<ShardedInputs>& in, unsigned log_min_sharded);
};

}  // namespace vhost

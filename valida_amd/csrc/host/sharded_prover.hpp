// ONE proof over W GPUs (intra-proof sharding): Machine::prove (basic/src/lib.rs:147-675) with every committed
// LDE, every Merkle tree, the quotient evaluation, the openings and the FRI commit phase held and computed in ROW-RANGE shards, one
// per rank, and the very proof words the single-GPU prover produces.  Design: sharded_prover.cpp.
#pragma once
#include "fabric.hpp"
#include "prover.hpp"

namespace vhost {

// The traces of one rank.  They are REPLICATED: every rank holds the whole main / preprocessed traces (the large objects of a
// proof — LDEs, trees, reduced openings, FRI layers — are what is sharded).
struct ShardedInputs {
    std::vector<const DeviceTrace*> main;
    std::vector<std::pair<int, const DeviceTrace*>> prep;
};

struct ShardedProof {
    // provers[k] / in[k]: the prover context and traces of hosted rank fabric.hosted[k].  Matrices whose LDE has at least
    // max(4 W, 2^log_min_sharded) rows are sharded, shorter ones are computed by every rank.  Returns the proof words (the same on every rank).
    // Multi-process fabrics: if this rank throws, its peers are told (Fabric::fail) and throw FabricPeerFailure at their next collective.
    static std::vector<uint32_t> run(Fabric& fabric, const std::vector<Prover*>& provers, const std::vector<ShardedInputs>& in, unsigned log_min_sharded);

  private:
    static std::vector<uint32_t> run_impl(Fabric& fabric, const std::vector<Prover*>& provers, const std::vector<ShardedInputs>& in, unsigned log_min_sharded);
};

}  // namespace vhost

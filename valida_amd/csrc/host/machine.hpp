// Host-side machine description: the compiled form of `[Box<&dyn Chip<Self, SC>>; NUM_CHIPS]`
// (basic/src/lib.rs:151-166).  Each chip contributes its width, its constraint program (compiled once
// from `Air::eval`, air/symbolic.hpp), its interactions and its log_quotient_degree
// (get_log_quotient_degree, machine/src/symbolic/symbolic_builder.rs:17-30).
#pragma once
#include <string>
#include "../air/symbolic.hpp"
#include "../chips/basic_machine.hpp"
#include "../kernels/interactions.hpp"

namespace vhost {

struct AirDesc {
    std::string name;
    uint32_t width = 0, prep_width = 0;
    vair::Program program;
    std::vector<vair::Interaction> interactions;
    std::vector<uint32_t> interaction_words;  // vk::encode_interactions
    unsigned log_quotient_degree = 1;
    int max_constraint_degree = 0;
    // vchips::ChipId when this AIR is one of the in-tree BasicMachine chips, whose eval template is also compiled
    // into a native quotient kernel (kernels/quotient.hip); -2 = interpret `program` (run-time captured AIRs)
    int native_chip = -2;
};

struct MachineDesc {
    std::vector<AirDesc> airs;

    // Build an AirDesc from an eval functor `void(Builder&)` instantiable with SymbolicBuilder and DegreeBuilder.
    template <class EvalSym, class EvalDeg>
    static AirDesc make_air(const std::string& name, uint32_t width, uint32_t prep_width, EvalSym eval_sym, EvalDeg eval_deg,
                            std::vector<vair::Interaction> interactions) {
        AirDesc a;
        a.name = name; a.width = width; a.prep_width = prep_width;
        vair::Dag dag;
        dag.width = (int)width; dag.prep_width = (int)prep_width;
        vair::SymbolicBuilder sb(&dag);
        eval_sym(sb);
        a.program = vair::compile(dag);
        vair::DegreeBuilder db;
        eval_deg(db);
        a.max_constraint_degree = db.max_degree;
        a.log_quotient_degree = vair::log_quotient_degree_from(db.max_degree);
        a.interactions = std::move(interactions);
        a.interaction_words = vk::encode_interactions(a.interactions);
        return a;
    }

    // From an externally captured DAG (the vgpu_air_* FFI path).  Degree comes from the DAG itself.
    static AirDesc make_air_from_dag(const std::string& name, const vair::Dag& dag, std::vector<vair::Interaction> interactions) {
        AirDesc a;
        a.name = name; a.width = (uint32_t)dag.width; a.prep_width = (uint32_t)dag.prep_width;
        a.program = vair::compile(dag);
        a.max_constraint_degree = dag.max_degree();
        a.log_quotient_degree = vair::log_quotient_degree_from(a.max_constraint_degree);
        a.interactions = std::move(interactions);
        a.interaction_words = vk::encode_interactions(a.interactions);
        return a;
    }

    static MachineDesc basic() {
        MachineDesc m;
        for (int i = 0; i < vchips::NUM_CHIPS; i++) {
            const auto& info = vchips::chip_info(i);
            m.airs.push_back(make_air(
                info.name, (uint32_t)info.width, (uint32_t)info.preprocessed_width, [i](vair::SymbolicBuilder& b) { vchips::eval_chip(i, b); },
                [i](vair::DegreeBuilder& b) { vchips::eval_chip(i, b); }, vchips::chip_interactions(i)));
            m.airs.back().native_chip = i;
        }
        return m;
    }
};

}  // namespace vhost

// adapter/fuse_pass.cpp -- see fuse_pass.hpp.  The recogniser of openvino_tokenizers_amd/pipeline.py fuse(), over ov::Node chains.
#include "fuse_pass.hpp"

#include <openvino/core/extension.hpp>

#include "ops.hpp"

namespace ovtk_adapter {
namespace {

// outputs [0, n) of `producer` feed inputs [at, at + n) of `consumer`, in order
bool feeds(const std::shared_ptr<ov::Node>& producer, const std::shared_ptr<ov::Node>& consumer, size_t at, size_t n) {
    for (size_t k = 0; k < n; ++k) {
        const ov::Output<ov::Node> src = consumer->input_value(at + k);
        if (src.get_node_shared_ptr() != producer || src.get_index() != k) return false;
    }
    return true;
}
// every consumer of every output of `node` is `only`
bool feeds_nothing_else(const std::shared_ptr<ov::Node>& node, const ov::Node* only) {
    for (size_t k = 0; k < node->get_output_size(); ++k)
        for (const ov::Input<ov::Node>& in : node->output(k).get_target_inputs())
            if (in.get_node() != only) return false;
    return true;
}
void replace_outputs(const std::shared_ptr<ov::Node>& old_node, const std::shared_ptr<ov::Node>& new_node, size_t n) {
    for (size_t k = 0; k < n; ++k) old_node->output(k).replace(new_node->output(k));
    new_node->set_friendly_name(old_node->get_friendly_name());
    ov::copy_runtime_info(old_node, new_node);
}

// [SpecialTokensSplit ->] RegexSplit -> BPETokenizer
bool fuse_bpe(const std::shared_ptr<ov::Node>& node) {
    auto bpe = ov::as_type_ptr<BPETokenizer>(node);
    if (!bpe) return false;
    auto split = ov::as_type_ptr<RegexSplit>(bpe->input_value(0).get_node_shared_ptr());
    if (!split || !feeds(split, bpe, 0, 5) || !feeds_nothing_else(split, bpe.get())) return false;
    const size_t split_inputs = split->get_input_size();
    if (split_inputs != 6 && split_inputs != 7) return false;   // (the legacy 9-input form keeps its own node)
    const bool split_skips = split_inputs == 7;
    // what feeds the split: a SpecialTokensSplit (its six outputs in order), or the graph's strings
    std::shared_ptr<SpecialTokensSplit> special;
    if (split_skips) {
        special = ov::as_type_ptr<SpecialTokensSplit>(split->input_value(0).get_node_shared_ptr());
        if (special && !(feeds(special, split, 0, 6) && feeds_nothing_else(special, split.get()))) special = nullptr;
    }
    const std::shared_ptr<ov::Node> head = special ? std::static_pointer_cast<ov::Node>(special) : std::static_pointer_cast<ov::Node>(split);
    const bool has_skips = head->get_input_size() == 7;
    ov::OutputVector args;
    for (size_t k = 0; k < 5 + (has_skips ? 1u : 0u); ++k) args.push_back(head->input_value(k));
    if (special) args.push_back(special->input_value(5 + (has_skips ? 1u : 0u)));          // the special-tokens pattern
    args.push_back(split->input_value(split_inputs - 1));                                  // the split pattern
    for (size_t k = 5; k < bpe->get_input_size(); ++k) args.push_back(bpe->input_value(k));   // vocabulary, merges, added tokens
    auto fused = std::make_shared<FusedSplitBPE>(args, special, split, bpe, has_skips, bpe->get_input_size());
    replace_outputs(bpe, fused, 3);
    return true;
}

// RegexSplit(\s+, remove) -> RegexSplit(delimiters, isolate) -> WordpieceTokenizer: the library checks the two patterns itself
// (ovtk_wordpiece_encode_run: OVTK_E_UNSUPPORTED for any other pair), so the pass only takes the shape
bool fuse_wordpiece(const std::shared_ptr<ov::Node>& node) {
    auto wp = ov::as_type_ptr<WordpieceTokenizer>(node);
    if (!wp) return false;
    auto pu = ov::as_type_ptr<RegexSplit>(wp->input_value(0).get_node_shared_ptr());
    if (!pu || pu->get_input_size() != 6 || !feeds(pu, wp, 0, 5) || !feeds_nothing_else(pu, wp.get())) return false;
    auto ws = ov::as_type_ptr<RegexSplit>(pu->input_value(0).get_node_shared_ptr());
    if (!ws || ws->get_input_size() != 6 || !feeds(ws, pu, 0, 5) || !feeds_nothing_else(ws, pu.get())) return false;
    ov::OutputVector args;
    for (size_t k = 0; k < 5; ++k) args.push_back(ws->input_value(k));
    args.push_back(ws->input_value(5));
    args.push_back(pu->input_value(5));
    for (size_t k = 5; k < 9; ++k) args.push_back(wp->input_value(k));
    auto fused = std::make_shared<FusedSplitWordpiece>(args, ws, pu, wp);
    replace_outputs(wp, fused, 3);
    return true;
}

// VocabDecoder -> [ByteFallback] -> FuzeRagged (+ whoever reads the chars tensor behind it)
bool fuse_detokenizer(const std::shared_ptr<ov::Node>& node) {
    auto fuze = ov::as_type_ptr<FuzeRagged>(node);
    if (!fuze) return false;
    std::shared_ptr<ov::Node> strings = fuze->input_value(2).get_node_shared_ptr();   // begins / ends of the token strings
    auto fallback = ov::as_type_ptr<ByteFallback>(strings);
    std::shared_ptr<ov::Node> below = fallback ? fallback->input_value(0).get_node_shared_ptr() : strings;
    auto dec = ov::as_type_ptr<VocabDecoder>(below);
    if (!dec) return false;
    // wiring: fuze(ragged_begins, ragged_ends, begins, ends) <- dec outputs 0, 1 and the string tensor's begins / ends
    if (fuze->input_value(0).get_node_shared_ptr() != dec || fuze->input_value(0).get_index() != 0 || fuze->input_value(1).get_node_shared_ptr() != dec ||
        fuze->input_value(1).get_index() != 1)
        return false;
    if (fallback) {
        for (size_t k = 0; k < 3; ++k)
            if (fallback->input_value(k).get_node_shared_ptr() != dec || fallback->input_value(k).get_index() != 2 + k) return false;
        if (fuze->input_value(2).get_index() != 0 || fuze->input_value(3).get_node_shared_ptr() != fallback || fuze->input_value(3).get_index() != 1) return false;
    } else if (fuze->input_value(2).get_index() != 2 || fuze->input_value(3).get_node_shared_ptr() != dec || fuze->input_value(3).get_index() != 3) {
        return false;
    }
    ov::OutputVector args;
    for (size_t k = 0; k < dec->get_input_size(); ++k) args.push_back(dec->input_value(k));
    auto fused = std::make_shared<FusedDetokenize>(args, dec, fallback != nullptr);
    fuze->output(0).replace(fused->output(0));
    fuze->output(1).replace(fused->output(1));
    // the chars tensor: output 2 of ByteFallback, or output 4 of VocabDecoder
    (fallback ? fallback->output(2) : dec->output(4)).replace(fused->output(2));
    fused->set_friendly_name(fuze->get_friendly_name());
    ov::copy_runtime_info(fuze, fused);
    return true;
}

}  // namespace

bool FuseTokenizerChains::run_on_model(const std::shared_ptr<ov::Model>& model) {
    bool changed = false;
    for (const std::shared_ptr<ov::Node>& node : model->get_ordered_ops()) changed = fuse_bpe(node) || fuse_wordpiece(node) || fuse_detokenizer(node) || changed;
    return changed;
}

}  // namespace ovtk_adapter

namespace ov {
namespace tokenizers {
OPENVINO_API_C(bool) fuse_tokenizer_chains(const std::shared_ptr<ov::Model>& model) { return ovtk_adapter::FuseTokenizerChains().run_on_model(model); }
}  // namespace tokenizers
}  // namespace ov

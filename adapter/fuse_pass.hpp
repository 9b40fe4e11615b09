// adapter/fuse_pass.hpp -- the graph rewrite that puts the library's fused entry points behind the reference's operator interface.
//
// The reference builds a converted tokenizer as a chain of custom-op nodes, one evaluate() each
// (python/openvino_tokenizers/tokenizer_pipeline.py:1613-1636 for byte-level BPE, :392-435 + :641-659 for BERT, :1321-1371 for the
// detokenizer).  With the op classes of ops.hpp alone every node stages its tensors over PCIe both ways and runs its own kernels;
// this pass recognises the chains the library has ONE call for and replaces them by the fused nodes of ops.hpp:
//   [SpecialTokensSplit ->] RegexSplit -> BPETokenizer                                    =>  OvtkFusedSplitBPE       (ovtk_encode_run / _special_run)
//   RegexSplit(\s+, remove) -> RegexSplit(BERT delimiters, isolate) -> WordpieceTokenizer  =>  OvtkFusedSplitWordpiece (ovtk_wordpiece_encode_run)
//   VocabDecoder -> [ByteFallback] -> FuzeRagged                                           =>  OvtkFusedDetokenize     (ovtk_detokenize_run)
// A chain is taken only when its inner tensors have no other consumer.  The same recogniser over the Python mirror classes is
// openvino_tokenizers_amd/pipeline.py fuse(); THAT one runs in the tests (tests/test_pipeline_fuse.py) -- this file is parsed against the
// repository's mock of the OpenVINO headers only (no OpenVINO in the image: tests/adapter_syntax/README.md).
// Use: after core.read_model() / convert_tokenizer() and before compile_model():
//   ov::pass::Manager m; m.register_pass<ovtk_adapter::FuseTokenizerChains>(); m.run_passes(model);
// or the C entry point ov::tokenizers::fuse_tokenizer_chains(model).
#pragma once

#include <memory>

#include <openvino/core/extension.hpp>
#include <openvino/core/model.hpp>
#include <openvino/pass/pass.hpp>

namespace ovtk_adapter {

class FuseTokenizerChains : public ov::pass::ModelPass {
public:
    OPENVINO_MODEL_PASS_RTTI("ovtk_adapter::FuseTokenizerChains");
    bool run_on_model(const std::shared_ptr<ov::Model>& model) override;
};

}  // namespace ovtk_adapter

namespace ov {
namespace tokenizers {
// true when something was rewritten
OPENVINO_API_C(bool) fuse_tokenizer_chains(const std::shared_ptr<ov::Model>& model);
}  // namespace tokenizers
}  // namespace ov

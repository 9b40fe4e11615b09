// adapter/extension.cpp -- the two entry points OpenVINO and OpenVINO GenAI look for in a tokenizers extension library:
//   create_extensions()      emitted by OPENVINO_CREATE_EXTENSIONS, found by ov::Core::add_extension(path)
//                            (reference: src/ov_extension.cpp:72-109)
//   create_tokenizer_node()  resolved with dlsym by GenAI when it builds a tokenizer from a GGUF file; "signature must
//                            not be changed" (reference: src/tokenizers_factory.hpp:21-33, .cpp:23-74)
// The list below is the hot path of SURVEY.md section 8 (rows a1-a10, f1-f4).  Ops outside it (normalizers, SentencePiece,
// Unigram, the TensorFlow / ONNX conversion extensions ...) are not replaced: load the stock libopenvino_tokenizers.so
// next to this library -- the last add_extension() wins for a type name, so add this one second.
#include <openvino/core/extension.hpp>
#include <openvino/core/op_extension.hpp>
#include <openvino/core/any.hpp>

#include <cstdlib>

#include "ops.hpp"

using namespace ovtk_adapter;

// clang-format off
OPENVINO_CREATE_EXTENSIONS(
    std::vector<ov::Extension::Ptr>({
        std::make_shared<ov::OpExtension<RegexSplit>>(),
        std::make_shared<ov::OpExtension<BPETokenizer>>(),
        std::make_shared<ov::OpExtension<WordpieceTokenizer>>(),
        std::make_shared<ov::OpExtension<VocabEncoder>>(),
        std::make_shared<ov::OpExtension<RaggedToDense>>(),
        std::make_shared<ov::OpExtension<VocabDecoder>>(),
        std::make_shared<ov::OpExtension<ByteFallback>>(),
        std::make_shared<ov::OpExtension<FuzeRagged>>(),
        std::make_shared<ov::OpExtension<SpecialTokensSplit>>(),
        std::make_shared<ov::OpExtension<StringTensorUnpack>>(),
        std::make_shared<ov::OpExtension<StringTensorPack>>(),
        std::make_shared<ov::OpExtension<Truncate>>(),
        std::make_shared<ov::OpExtension<CombineSegments>>(),
        std::make_shared<ov::OpExtension<UTF8Validate>>(),
        std::make_shared<ov::OpExtension<TrieTokenizer>>(),
    }));
// clang-format on

namespace {
// Device memory per tokenizer node: every BPETokenizer / WordpieceTokenizer handle owns a piece store sized by the library's default
// (include/ovtk_amd.h ovtk_set_memo_store: GPT-2 64 MB, Llama-3 128 MB) -- the nodes are created with memo_store = 0 --, and a graph that
// is cloned per infer request multiplies it.  OVTK_AMD_MEMO_STORE=<entries> (0 or less: no store) sets that default once, when this
// library is loaded; it changes no result, only what a repeated text costs.
const bool memo_store_from_env = [] {
    if (const char* e = std::getenv("OVTK_AMD_MEMO_STORE")) {
        const long long n = std::atoll(e);
        ovtk_set_memo_store(n > 0 ? n : -1);
    }
    return true;
}();

template <typename T>
T attr(const ov::AnyMap& attributes, const std::string& name, const T& fallback) {
    auto it = attributes.find(name);
    return it != attributes.end() && it->second.is<T>() ? it->second.as<T>() : fallback;
}
}  // namespace

namespace ov {
namespace tokenizers {

// Same name, linkage and signature as the reference's factory; the op types of the hot path only.
OPENVINO_API_C(ov::OutputVector)
create_tokenizer_node(const std::string& op_type, const ov::OutputVector& inputs, const ov::AnyMap& attributes) {
    if (op_type == "StringTensorUnpack") return std::make_shared<StringTensorUnpack>(inputs)->outputs();
    if (op_type == "StringTensorPack") return std::make_shared<StringTensorPack>(inputs)->outputs();
    if (op_type == "SpecialTokensSplit") return std::make_shared<SpecialTokensSplit>(inputs)->outputs();
    if (op_type == "RegexSplit")
        return std::make_shared<RegexSplit>(inputs, attr<std::string>(attributes, "behaviour", "remove"), attr<bool>(attributes, "invert", false),
                                            attr<int>(attributes, "max_splits", -1))
            ->outputs();
    if (op_type == "BPETokenizer")
        return std::make_shared<BPETokenizer>(inputs, attr<std::string>(attributes, "unk_token", ""), attr<bool>(attributes, "fuse_unk", false),
                                              attr<std::string>(attributes, "suffix_indicator", ""), attr<std::string>(attributes, "end_suffix", ""),
                                              attr<bool>(attributes, "byte_fallback", false))
            ->outputs();
    if (op_type == "WordpieceTokenizer")
        return std::make_shared<WordpieceTokenizer>(inputs, attr<std::string>(attributes, "suffix_indicator", "##"),
                                                    attr<int>(attributes, "max_bytes_per_word", 100))
            ->outputs();
    if (op_type == "VocabEncoder") return std::make_shared<VocabEncoder>(inputs)->outputs();
    if (op_type == "RaggedToDense")
        return std::make_shared<RaggedToDense>(inputs, attr<bool>(attributes, "pad_right", true), attr<bool>(attributes, "pad_max_length", false))
            ->outputs();
    if (op_type == "VocabDecoder") return std::make_shared<VocabDecoder>(inputs, std::vector<int>{})->outputs();
    if (op_type == "ByteFallback") return std::make_shared<ByteFallback>(inputs)->outputs();
    if (op_type == "FuzeRagged") return std::make_shared<FuzeRagged>(inputs)->outputs();
    if (op_type == "Truncate") return std::make_shared<Truncate>(inputs)->outputs();
    if (op_type == "CombineSegments") return std::make_shared<CombineSegments>(inputs)->outputs();
    if (op_type == "UTF8Validate") return std::make_shared<UTF8Validate>(inputs, attr<bool>(attributes, "replace_mode", false))->outputs();
    if (op_type == "TrieTokenizer") return std::make_shared<TrieTokenizer>(inputs)->outputs();
    OPENVINO_THROW("Unsupported operation type: `", op_type, "` (this library replaces the tokenizer hot path only)");
}

}  // namespace tokenizers
}  // namespace ov

// adapter/ops.hpp -- the OpenVINO custom ops of the tokenizer hot path, backed by libovtk_amd.so.
//
// Each class carries the reference op's type name (OPENVINO_OP), constructor arguments, attribute names
// (visit_attributes = the IR / NodeFactory names) and input / output order, so an IR or a pipeline built for
// openvino_tokenizers loads unchanged; evaluate() hands the host tensors to the C ABI (OVTK_MEM_HOST: the library
// stages them over PCIe) and shrinks the variable-length outputs to what was produced.  Reference classes replaced:
//   RegexSplit            src/regex_split.hpp / .cpp           BPETokenizer      src/bpe_tokenizer.hpp / .cpp
//   WordpieceTokenizer    src/wordpiece_tokenizer.hpp / .cpp   VocabEncoder      src/vocab_encoder.hpp / .cpp
//   RaggedToDense         src/ragged_to_dense.hpp / .cpp       VocabDecoder      src/vocab_decoder.hpp / .cpp
//   ByteFallback          src/byte_fallback.hpp / .cpp         FuzeRagged        src/fuze.hpp / .cpp
//   SpecialTokensSplit    src/special_tokens_split.hpp / .cpp  Truncate          src/truncate.hpp / .cpp
//   CombineSegments       src/combine_segments.hpp / .cpp      UTF8Validate      src/utf8_validate.hpp / .cpp
//   TrieTokenizer         src/trie_tokenizer.hpp / .cpp        StringTensorUnpack (u8 wire form) / StringTensorPack
// Device tables are built on the first evaluate() under a mutex and shared by clones, like the reference's lazily built
// state (bpe_tokenizer.hpp:215-218, regex_split.hpp:37-40).  evaluate() is const and re-entrant: the handles are
// immutable, every call leases its own workspace inside the library.
#pragma once

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <openvino/op/op.hpp>
#include <openvino/op/string_tensor_pack.hpp>

#include "ovtk_amd.h"

namespace ovtk_adapter {

// A library handle created once, destroyed with the last node that shares it.
template <class H>
struct Lazy {
    std::mutex mutex;
    H* handle = nullptr;
    void (*destroy)(H*) = nullptr;
    ~Lazy() {
        if (handle && destroy) destroy(handle);
    }
};

// What every op below has in common: evaluate() exists, the device ordinal the tables live on.
class Base : public ov::op::Op {
public:
    using ov::op::Op::Op;
    bool has_evaluate() const override { return true; }

protected:
    static int device();  // OVTK_DEVICE environment variable, default 0
};

class RegexSplit : public Base {
public:
    OPENVINO_OP("RegexSplit");
    RegexSplit() = default;
    RegexSplit(const ov::OutputVector& arguments, const std::string& behaviour = "remove", bool invert = false, int max_splits = -1);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;
    // the device tables of this node (built on first use, shared by its clones): for the fused nodes of fuse_pass.hpp
    ovtk_regex_split* handle(const ov::Tensor& pattern) const;

private:
    std::string m_behaviour = "remove";
    bool m_invert = false;
    int m_max_splits = -1;
    mutable std::shared_ptr<Lazy<ovtk_regex_split>> m_state = std::make_shared<Lazy<ovtk_regex_split>>();
    // the 9-input form of old IRs: the set of "skip tokens" (inputs 6-8) as a string -> flag map on the device
    mutable std::shared_ptr<Lazy<ovtk_vocab_encoder>> m_skip_set = std::make_shared<Lazy<ovtk_vocab_encoder>>();
};

class SpecialTokensSplit : public Base {
public:
    OPENVINO_OP("SpecialTokensSplit");
    SpecialTokensSplit() = default;
    explicit SpecialTokensSplit(const ov::OutputVector& arguments);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor&) override { return true; }
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;
    ovtk_special_tokens_split* handle(const ov::Tensor& pattern) const;

private:
    mutable std::shared_ptr<Lazy<ovtk_special_tokens_split>> m_state = std::make_shared<Lazy<ovtk_special_tokens_split>>();
};

class BPETokenizer : public Base {
public:
    OPENVINO_OP("BPETokenizer");
    BPETokenizer() = default;
    BPETokenizer(const ov::OutputVector& arguments, const std::string& unk_token = "", bool fuse_unk = false,
                 const std::string& suffix_indicator = "", const std::string& end_suffix = "", bool byte_fallback = false,
                 int64_t cache_capacity = 20000);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;
    // `consts`: inputs 5.. of the op (vocab, merges, [added tokens]), `first`: where they start in `consts`
    ovtk_bpe* handle(const ov::TensorVector& consts, size_t first, size_t n_inputs) const;
    size_t end_suffix_size() const { return m_end_suffix.size(); }

private:
    std::string m_unk_token, m_suffix_indicator, m_end_suffix;
    bool m_fuse_unk = false, m_byte_fallback = false;
    int64_t m_cache_capacity = 20000;
    mutable std::shared_ptr<Lazy<ovtk_bpe>> m_state = std::make_shared<Lazy<ovtk_bpe>>();
};

class WordpieceTokenizer : public Base {
public:
    OPENVINO_OP("WordpieceTokenizer");
    WordpieceTokenizer() = default;
    WordpieceTokenizer(const ov::OutputVector& arguments, const std::string& suffix_indicator = "##", int max_bytes_per_word = 100);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;
    ovtk_wordpiece* handle(const ov::TensorVector& tensors, size_t vocab_at) const;

private:
    std::string m_suffix_indicator = "##";
    int m_max_bytes_per_word = 100;
    mutable std::shared_ptr<Lazy<ovtk_wordpiece>> m_state = std::make_shared<Lazy<ovtk_wordpiece>>();
};

class VocabEncoder : public Base {
public:
    OPENVINO_OP("VocabEncoder");
    VocabEncoder() = default;
    explicit VocabEncoder(const ov::OutputVector& arguments);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor&) override { return true; }
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    mutable std::shared_ptr<Lazy<ovtk_vocab_encoder>> m_state = std::make_shared<Lazy<ovtk_vocab_encoder>>();
};

class RaggedToDense : public Base {
public:
    OPENVINO_OP("RaggedToDense");
    RaggedToDense() = default;
    RaggedToDense(const ov::OutputVector& arguments, bool pad_right = true, bool pad_max_length = false);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    bool m_pad_right = true, m_pad_max_length = false;
};

class VocabDecoder : public Base {
public:
    OPENVINO_OP("VocabDecoder");
    VocabDecoder() = default;
    VocabDecoder(const ov::OutputVector& arguments, std::vector<int> skip_tokens = {});
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;
    ovtk_vocab_decoder* handle(const ov::TensorVector& inputs) const;

private:
    std::vector<int> m_skip_tokens;
    mutable std::shared_ptr<Lazy<ovtk_vocab_decoder>> m_state = std::make_shared<Lazy<ovtk_vocab_decoder>>();
};

// The stateless ops: one forwarding evaluate() each.
#define OVTK_ADAPTER_STATELESS_OP(Name)                                                                       \
    class Name : public Base {                                                                                \
    public:                                                                                                   \
        OPENVINO_OP(#Name);                                                                                   \
        Name() = default;                                                                                     \
        explicit Name(const ov::OutputVector& arguments) : Base(arguments) { constructor_validate_and_infer_types(); } \
        void validate_and_infer_types() override;                                                             \
        std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override {      \
            return std::make_shared<Name>(inputs);                                                            \
        }                                                                                                     \
        bool visit_attributes(ov::AttributeVisitor&) override { return true; }                                \
        bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;             \
    }
OVTK_ADAPTER_STATELESS_OP(ByteFallback);
OVTK_ADAPTER_STATELESS_OP(FuzeRagged);
OVTK_ADAPTER_STATELESS_OP(CombineSegments);
#undef OVTK_ADAPTER_STATELESS_OP

class Truncate : public Base {
public:
    OPENVINO_OP("Truncate");
    Truncate() = default;
    explicit Truncate(const ov::OutputVector& arguments);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    int64_t m_num_inputs = 1;  // ragged tensors truncated together: 1 or 2 (truncate.hpp: "m_num_inputs")
};

class UTF8Validate : public Base {
public:
    OPENVINO_OP("UTF8Validate");
    UTF8Validate() = default;
    UTF8Validate(const ov::OutputVector& arguments, bool replace_mode = false);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    bool m_replace_mode = false;
};

class TrieTokenizer : public Base {
public:
    OPENVINO_OP("TrieTokenizer");
    TrieTokenizer() = default;
    explicit TrieTokenizer(const ov::OutputVector& arguments);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor&) override { return true; }
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    mutable std::shared_ptr<Lazy<ovtk_trie_tokenizer>> m_state = std::make_shared<Lazy<ovtk_trie_tokenizer>>();
};

// The packed-u8 branch of StringTensorUnpack / StringTensorPack (string_tensor_unpack.cpp:53-71, utils.cpp:18-29): the
// wire form [i32 n][i32 begin_0][i32 end_i x n][bytes].  The element::string branch holds std::string objects, a host
// container with nothing for a GPU to do: not registered here (the stock op keeps serving it).
class StringTensorUnpack : public Base {
public:
    OPENVINO_OP("StringTensorUnpack");
    StringTensorUnpack() = default;
    StringTensorUnpack(const ov::OutputVector& arguments, const std::string& mode = "begins_ends");
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    std::string m_mode = "begins_ends";
};

// StringTensorPack of the "extension" opset (src/string_tensor_pack.hpp:13-43; extension list src/ov_extension.cpp:74,
// factory src/tokenizers_factory.cpp:41-42): in the reference it is nothing but the official opset-15 op under the old
// type name, kept for IRs written before opset 15 -- its output is an element::string tensor, i.e. std::string objects on
// the host, and its evaluate() is OpenVINO's own.  Same here: the class exists so that such IRs and GenAI's
// create_tokenizer_node("StringTensorPack") resolve against this library alone; there is nothing for the GPU in it.  (The
// packed-u8 wire form, the one that crosses PCIe, is ovtk_string_tensor_pack in the C ABI.)
class StringTensorPack : public ov::op::v15::StringTensorPack {
public:
    OPENVINO_OP("StringTensorPack", "extension", ov::op::v15::StringTensorPack);
    StringTensorPack() = default;
    StringTensorPack(const ov::OutputVector& arguments, const std::string& mode = "begins_ends");
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor& visitor) override;

private:
    std::string m_mode = "begins_ends";
};

// ---- the fused nodes fuse_pass.cpp puts in place of the chains it recognises (python mirror: openvino_tokenizers_amd/pipeline.py) ------
// Each holds the nodes it replaces -- their attributes and their device tables -- and forwards ONE evaluate() to the library's fused
// entry point, host tensors in and out (OVTK_MEM_HOST: staged over PCIe once per chain instead of once per op).

// [SpecialTokensSplit ->] RegexSplit -> BPETokenizer  =>  ovtk_encode_run / ovtk_encode_special_run.
// Inputs: the chain's ragged strings 0-4 [, skips], [the special pattern], the split pattern, then BPETokenizer's inputs 5...
class FusedSplitBPE : public Base {
public:
    OPENVINO_OP("OvtkFusedSplitBPE");
    FusedSplitBPE() = default;
    FusedSplitBPE(const ov::OutputVector& arguments, std::shared_ptr<const SpecialTokensSplit> special, std::shared_ptr<const RegexSplit> split,
                  std::shared_ptr<const BPETokenizer> bpe, bool has_skips, size_t bpe_inputs);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor&) override { return true; }
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    std::shared_ptr<const SpecialTokensSplit> m_special;
    std::shared_ptr<const RegexSplit> m_split;
    std::shared_ptr<const BPETokenizer> m_bpe;
    bool m_has_skips = false;
    size_t m_bpe_inputs = 0;   // input count of the BPETokenizer node (11 / 14 / 15 / 18: which constants follow)
};

// RegexSplit(\\s+, remove) -> RegexSplit(BERT delimiters, isolate) -> WordpieceTokenizer  =>  ovtk_wordpiece_encode_run.
// Inputs: ragged strings 0-4, the two patterns, the vocabulary (3), unk_token_id.
class FusedSplitWordpiece : public Base {
public:
    OPENVINO_OP("OvtkFusedSplitWordpiece");
    FusedSplitWordpiece() = default;
    FusedSplitWordpiece(const ov::OutputVector& arguments, std::shared_ptr<const RegexSplit> whitespace, std::shared_ptr<const RegexSplit> delimiters,
                        std::shared_ptr<const WordpieceTokenizer> wordpiece);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor&) override { return true; }
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    std::shared_ptr<const RegexSplit> m_whitespace, m_delimiters;
    std::shared_ptr<const WordpieceTokenizer> m_wordpiece;
};

// VocabDecoder -> [ByteFallback] -> FuzeRagged  =>  ovtk_detokenize_run.  Inputs: VocabDecoder's; outputs: begins, ends, chars of the rows.
class FusedDetokenize : public Base {
public:
    OPENVINO_OP("OvtkFusedDetokenize");
    FusedDetokenize() = default;
    FusedDetokenize(const ov::OutputVector& arguments, std::shared_ptr<const VocabDecoder> decoder, bool byte_fallback);
    void validate_and_infer_types() override;
    std::shared_ptr<ov::Node> clone_with_new_inputs(const ov::OutputVector& inputs) const override;
    bool visit_attributes(ov::AttributeVisitor&) override { return true; }
    bool evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const override;

private:
    std::shared_ptr<const VocabDecoder> m_decoder;
    bool m_byte_fallback = false;
};

}  // namespace ovtk_adapter

// adapter/ops.cpp -- see ops.hpp.  Compiled only where the OpenVINO developer package exists (adapter/CMakeLists.txt).
#include "ops.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <openvino/core/except.hpp>
#include <openvino/runtime/tensor.hpp>

namespace ovtk_adapter {
namespace {

using ov::element::Type;
namespace el = ov::element;

// No exception crosses the C ABI: a failed call becomes ov::Exception here, with the library's message.
void check(int rc, const char* what) {
    if (rc != OVTK_OK) OPENVINO_THROW(what, " failed (ovtk error ", rc, "): ", ovtk_last_error());
}

const int32_t* i32(const ov::Tensor& t) { return static_cast<const int32_t*>(t.data()); }
int32_t* i32(ov::Tensor& t) { return static_cast<int32_t*>(t.data()); }
const uint8_t* u8(const ov::Tensor& t) { return static_cast<const uint8_t*>(t.data()); }
uint8_t* u8(ov::Tensor& t) { return static_cast<uint8_t*>(t.data()); }

// Decomposed string tensor at inputs[i .. i+2] (src/utils.cpp:84-88).
ovtk_strings strings_at(const ov::TensorVector& in, size_t i) {
    return ovtk_strings{i32(in[i]), i32(in[i + 1]), u8(in[i + 2]), int64_t(in[i].get_size()), int64_t(in[i + 2].get_size())};
}
// Decomposed ragged string tensor at inputs[0 .. 4] (src/utils.cpp:90-96).
ovtk_ragged_strings ragged_at(const ov::TensorVector& in) {
    return ovtk_ragged_strings{i32(in[0]), i32(in[1]), int64_t(in[0].get_size()), strings_at(in, 2)};
}
std::string text_of(const ov::Tensor& t) { return std::string(static_cast<const char*>(t.data()), t.get_size()); }

void expect_i32(const ov::Node* n, size_t i, const char* what) {
    OPENVINO_ASSERT(n->get_input_element_type(i) == el::i32 || n->get_input_element_type(i) == el::dynamic, what,
                    ": expected an i32 tensor at input ", i, ", got ", n->get_input_element_type(i));
}
void expect_strings(const ov::Node* n, size_t i, const char* what) {
    expect_i32(n, i, what);
    expect_i32(n, i + 1, what);
    OPENVINO_ASSERT(n->get_input_element_type(i + 2) == el::u8 || n->get_input_element_type(i + 2) == el::dynamic, what,
                    ": expected a u8 tensor at input ", i + 2);
}
void expect_ragged_strings(const ov::Node* n, const char* what) {
    expect_i32(n, 0, what);
    expect_i32(n, 1, what);
    expect_strings(n, 2, what);
}
const ov::PartialShape kDyn1D{ov::Dimension()};
void ragged_i32_outputs(ov::Node* n, const ov::PartialShape& rows) {  // begins, ends, flat i32 elements
    n->set_output_type(0, el::i32, rows);
    n->set_output_type(1, el::i32, rows);
    n->set_output_type(2, el::i32, kDyn1D);
}
void ragged_string_outputs(ov::Node* n, const ov::PartialShape& rows) {
    n->set_output_type(0, el::i32, rows);
    n->set_output_type(1, el::i32, rows);
    n->set_output_type(2, el::i32, kDyn1D);
    n->set_output_type(3, el::i32, kDyn1D);
    n->set_output_type(4, el::u8, kDyn1D);
}
void string_outputs(ov::Node* n, size_t at, const ov::PartialShape& shape) {
    n->set_output_type(at, el::i32, shape);
    n->set_output_type(at + 1, el::i32, shape);
    n->set_output_type(at + 2, el::u8, kDyn1D);
}

// The first evaluate() builds the device tables (the reference's call_once / mutex-guarded lazy init).
template <class H, class Create>
H* ensure(Lazy<H>& s, void (*destroy)(H*), Create&& create) {
    std::lock_guard<std::mutex> lock(s.mutex);
    if (!s.handle) {
        s.destroy = destroy;
        create(&s.handle);
    }
    return s.handle;
}

ov::Shape one_dim(size_t n) { return ov::Shape{n}; }

}  // namespace

int Base::device() {
    const char* v = std::getenv("OVTK_DEVICE");
    return v ? std::atoi(v) : 0;
}

// ------------------------------------------------------------------------------------------------ RegexSplit
RegexSplit::RegexSplit(const ov::OutputVector& arguments, const std::string& behaviour, bool invert, int max_splits)
    : Base(arguments), m_behaviour(behaviour), m_invert(invert), m_max_splits(max_splits) {
    constructor_validate_and_infer_types();
}
void RegexSplit::validate_and_infer_types() {
    const size_t n = get_input_size();
    OPENVINO_ASSERT(n == 6 || n == 7 || n == 9, "Incorrect number of inputs passed to RegexSplit: ", n,
                    "; try to reconvert tokenizer with newer version of OpenVINO Tokenizers");
    expect_ragged_strings(this, "RegexSplit");
    if (n == 9) expect_strings(this, 6, "RegexSplit skip tokens");
    OPENVINO_ASSERT(m_max_splits == -1 || m_max_splits > 0, "RegexSplit max_splits attribute must be greater then `0` or equal to `-1`, got ",
                    m_max_splits);
    ragged_string_outputs(this, get_input_partial_shape(0));
    if (n == 7) set_output_type(5, el::boolean, kDyn1D);
}
std::shared_ptr<ov::Node> RegexSplit::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<RegexSplit>(inputs, m_behaviour, m_invert, m_max_splits);
    c->m_state = m_state;  // the compiled pattern is shared (regex_split.hpp:37-40)
    c->m_skip_set = m_skip_set;
    return c;
}
bool RegexSplit::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("behaviour", m_behaviour);
    visitor.on_attribute("invert", m_invert);
    visitor.on_attribute("max_splits", m_max_splits);
    return true;
}
ovtk_regex_split* RegexSplit::handle(const ov::Tensor& pattern) const {
    return ensure(*m_state, ovtk_regex_split_destroy, [&](ovtk_regex_split** out) {
        const std::string pat = text_of(pattern);
        const ovtk_regex_split_params p{pat.data(), int64_t(pat.size()), m_behaviour.c_str(), m_invert ? 1 : 0, m_max_splits, device()};
        check(ovtk_regex_split_create(&p, out), "RegexSplit (pattern compilation)");
    });
}
bool RegexSplit::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const bool has_skips = inputs.size() == 7;
    // The 9-input form of old IRs (regex_split.cpp:102,164-179,235-238): a string that equals one of the "skip tokens"
    // (inputs 6-8) passes through unsplit.  Membership in that set is the VocabEncoder kernel with the tokens as keys (value
    // 1, default 0); the flags then play the part of the skips input, and do not become an output.
    std::vector<uint8_t> legacy_skips;
    if (inputs.size() == 9 && inputs[6].get_size() > 0 && inputs[4].get_size() > 0) {
        const std::vector<int32_t> ones(inputs[6].get_size(), 1);
        ovtk_vocab_encoder* set = ensure(*m_skip_set, ovtk_vocab_encoder_destroy, [&](ovtk_vocab_encoder** out) {
            const ovtk_vocab_encoder_params p{strings_at(inputs, 6), ones.data(), 4, device()};
            check(ovtk_vocab_encoder_create(&p, out), "RegexSplit (skip-token set)");
        });
        const ovtk_strings strings = strings_at(inputs, 2);
        std::vector<int32_t> flags(size_t(strings.n), 0);
        const int32_t none = 0;
        check(ovtk_vocab_encoder_run(set, &strings, &none, flags.data(), OVTK_MEM_HOST, nullptr), "RegexSplit (skip tokens)");
        legacy_skips.assign(flags.begin(), flags.end());
    }
    ovtk_regex_split* h = handle(inputs[5 + has_skips]);
    const ovtk_ragged_strings in = ragged_at(inputs);
    const size_t rows = inputs[0].get_size();
    const size_t cap = inputs[4].get_size() + inputs[2].get_size();  // regex_split.cpp:182
    outputs[0].set_shape(one_dim(std::max<size_t>(rows, 1)));
    outputs[1].set_shape(one_dim(std::max<size_t>(rows, 1)));
    outputs[2].set_shape(one_dim(cap));
    outputs[3].set_shape(one_dim(cap));
    if (has_skips) outputs[5].set_shape(one_dim(cap));
    std::vector<uint8_t> dropped_skips(legacy_skips.empty() ? 0 : cap);  // (the library writes a skips tensor whenever it reads one)
    ovtk_ragged_strings_out out{i32(outputs[0]), i32(outputs[1]), 0, i32(outputs[2]), i32(outputs[3]),
                                has_skips ? u8(outputs[5]) : (legacy_skips.empty() ? nullptr : dropped_skips.data()), int64_t(cap), 0};
    const uint8_t* skips_in = has_skips ? u8(inputs[5]) : (legacy_skips.empty() ? nullptr : legacy_skips.data());
    check(ovtk_regex_split_run(h, &in, skips_in, &out, OVTK_MEM_HOST, nullptr), "RegexSplit");
    outputs[0].set_shape(one_dim(size_t(out.n_rows)));
    outputs[1].set_shape(one_dim(size_t(out.n_rows)));
    if (out.n < 0) {  // the all-empty batch: the string tensors pass through (regex_split.cpp:129-143)
        outputs[2] = inputs[2];
        outputs[3] = inputs[3];
        if (has_skips) outputs[5] = inputs[5];
    } else {
        outputs[2].set_shape(one_dim(size_t(out.n)));
        outputs[3].set_shape(one_dim(size_t(out.n)));
        if (has_skips) outputs[5].set_shape(one_dim(size_t(out.n)));
    }
    outputs[4] = inputs[4];  // the chars tensor is the input's (regex_split.cpp:203)
    return true;
}

// ------------------------------------------------------------------------------------------------ SpecialTokensSplit
SpecialTokensSplit::SpecialTokensSplit(const ov::OutputVector& arguments) : Base(arguments) { constructor_validate_and_infer_types(); }
void SpecialTokensSplit::validate_and_infer_types() {
    const size_t n = get_input_size();
    OPENVINO_ASSERT(n == 6 || n == 7, "Incorrect number of inputs passed to SpecialTokensSplit: ", n,
                    "; try to reconvert tokenizer with newer version of OpenVINO Tokenizers");
    expect_ragged_strings(this, "SpecialTokensSplit");
    ragged_string_outputs(this, get_input_partial_shape(0));
    set_output_type(5, el::boolean, kDyn1D);
}
std::shared_ptr<ov::Node> SpecialTokensSplit::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<SpecialTokensSplit>(inputs);
    c->m_state = m_state;
    return c;
}
ovtk_special_tokens_split* SpecialTokensSplit::handle(const ov::Tensor& pattern) const {
    return ensure(*m_state, ovtk_special_tokens_split_destroy, [&](ovtk_special_tokens_split** out) {
        const std::string pat = text_of(pattern);
        check(ovtk_special_tokens_split_create(pat.data(), int64_t(pat.size()), device(), out), "SpecialTokensSplit (pattern)");
    });
}
bool SpecialTokensSplit::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const bool has_skips = inputs.size() == 7;
    ovtk_special_tokens_split* h = handle(inputs[5 + has_skips]);
    const ovtk_ragged_strings in = ragged_at(inputs);
    const size_t rows = inputs[0].get_size(), cap = inputs[4].get_size() + inputs[2].get_size();
    outputs[0].set_shape(one_dim(rows));
    outputs[1].set_shape(one_dim(rows));
    for (size_t k : {size_t(2), size_t(3), size_t(5)}) outputs[k].set_shape(one_dim(cap));
    ovtk_ragged_strings_out out{i32(outputs[0]), i32(outputs[1]), 0, i32(outputs[2]), i32(outputs[3]), u8(outputs[5]), int64_t(cap), 0};
    check(ovtk_special_tokens_split_run(h, &in, has_skips ? u8(inputs[5]) : nullptr, &out, OVTK_MEM_HOST, nullptr), "SpecialTokensSplit");
    for (size_t k : {size_t(2), size_t(3), size_t(5)}) outputs[k].set_shape(one_dim(size_t(out.n)));
    outputs[4] = inputs[4];
    return true;
}

// ------------------------------------------------------------------------------------------------ BPETokenizer
BPETokenizer::BPETokenizer(const ov::OutputVector& arguments, const std::string& unk_token, bool fuse_unk,
                           const std::string& suffix_indicator, const std::string& end_suffix, bool byte_fallback, int64_t cache_capacity)
    : Base(arguments), m_unk_token(unk_token), m_suffix_indicator(suffix_indicator), m_end_suffix(end_suffix), m_fuse_unk(fuse_unk),
      m_byte_fallback(byte_fallback), m_cache_capacity(cache_capacity) {
    constructor_validate_and_infer_types();
}
void BPETokenizer::validate_and_infer_types() {
    const size_t n = get_input_size();
    OPENVINO_ASSERT(n == 11 || n == 14 || n == 15 || n == 18, "Incorrect number of inputs passed to BPETokenizer, try to reconvert "
                                                             "tokenizer with newer version of OpenVINO Tokenizers");
    expect_ragged_strings(this, "BPETokenizer");
    expect_strings(this, 5, "BPETokenizer vocab");
    expect_strings(this, 8, "BPETokenizer merges");
    if (n == 14 || n == 18) expect_strings(this, 11, "BPETokenizer merges (right halves)");
    if (n == 15 || n == 18) {
        expect_strings(this, n - 4, "BPETokenizer added tokens");
        expect_i32(this, n - 1, "BPETokenizer added token ids");
    }
    ragged_i32_outputs(this, get_input_partial_shape(0));
}
std::shared_ptr<ov::Node> BPETokenizer::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<BPETokenizer>(inputs, m_unk_token, m_fuse_unk, m_suffix_indicator, m_end_suffix, m_byte_fallback, m_cache_capacity);
    c->m_state = m_state;  // vocabulary / merge tables are shared (bpe_tokenizer.hpp:215-218)
    return c;
}
bool BPETokenizer::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("unk_token", m_unk_token);
    visitor.on_attribute("fuse_unk", m_fuse_unk);
    visitor.on_attribute("suffix_indicator", m_suffix_indicator);
    visitor.on_attribute("end_suffix", m_end_suffix);
    visitor.on_attribute("byte_fallback", m_byte_fallback);
    visitor.on_attribute("cache_capacity", m_cache_capacity);
    return true;
}
ovtk_bpe* BPETokenizer::handle(const ov::TensorVector& t, size_t first, size_t n) const {
    // t[first ..] = the op's inputs 5 ..; n = the op's input count (which constants there are)
    return ensure(*m_state, ovtk_bpe_destroy, [&](ovtk_bpe** out) {
        ovtk_bpe_params p{};
        p.vocab = strings_at(t, first);
        p.merges = strings_at(t, first + 3);  // "left right" lines (11 / 15 inputs) or the left halves (14 / 18)
        if (n == 14 || n == 18) p.merges_right = strings_at(t, first + 6);
        if (n == 15 || n == 18) {
            p.added_tokens = strings_at(t, first + n - 9);
            p.added_ids = i32(t[first + n - 6]);
        }
        p.unk_token = m_unk_token.data();
        p.unk_token_len = int64_t(m_unk_token.size());
        p.fuse_unk = m_fuse_unk;
        p.suffix_indicator = m_suffix_indicator.data();
        p.suffix_indicator_len = int64_t(m_suffix_indicator.size());
        p.end_suffix = m_end_suffix.data();
        p.end_suffix_len = int64_t(m_end_suffix.size());
        p.byte_fallback = m_byte_fallback;
        p.cache_capacity = m_cache_capacity;
        p.device = device();
        check(ovtk_bpe_create(&p, out), "BPETokenizer (table construction)");
    });
}
bool BPETokenizer::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    ovtk_bpe* h = handle(inputs, 5, inputs.size());
    const ovtk_ragged_strings in = ragged_at(inputs);
    // the reference sizes the ids to the chars tensor (bpe_tokenizer.cpp:135); an end_suffix adds up to its length per piece
    const size_t cap = (inputs[4].get_size() + inputs[2].get_size()) * (1 + m_end_suffix.size());
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_ragged_i32_out out{i32(outputs[0]), i32(outputs[1]), i32(outputs[2]), int64_t(cap), 0, 0};
    check(ovtk_bpe_run(h, &in, &out, OVTK_MEM_HOST, nullptr), "BPETokenizer");
    outputs[2].set_shape(one_dim(size_t(out.n_data)));
    return true;
}

// ------------------------------------------------------------------------------------------------ WordpieceTokenizer
WordpieceTokenizer::WordpieceTokenizer(const ov::OutputVector& arguments, const std::string& suffix_indicator, int max_bytes_per_word)
    : Base(arguments), m_suffix_indicator(suffix_indicator), m_max_bytes_per_word(max_bytes_per_word) {
    constructor_validate_and_infer_types();
}
void WordpieceTokenizer::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 9, "Incorrect number of inputs passed to WordpieceTokenizer: ", get_input_size());
    expect_ragged_strings(this, "WordpieceTokenizer");
    expect_strings(this, 5, "WordpieceTokenizer vocab");
    ragged_i32_outputs(this, get_input_partial_shape(0));
}
std::shared_ptr<ov::Node> WordpieceTokenizer::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<WordpieceTokenizer>(inputs, m_suffix_indicator, m_max_bytes_per_word);
    c->m_state = m_state;
    return c;
}
bool WordpieceTokenizer::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("suffix_indicator", m_suffix_indicator);
    visitor.on_attribute("max_bytes_per_word", m_max_bytes_per_word);
    return true;
}
ovtk_wordpiece* WordpieceTokenizer::handle(const ov::TensorVector& t, size_t vocab_at) const {
    return ensure(*m_state, ovtk_wordpiece_destroy, [&](ovtk_wordpiece** out) {
        const ovtk_wordpiece_params p{strings_at(t, vocab_at), m_suffix_indicator.data(), int64_t(m_suffix_indicator.size()),
                                      m_max_bytes_per_word, device()};
        check(ovtk_wordpiece_create(&p, out), "WordpieceTokenizer (trie construction)");
    });
}
bool WordpieceTokenizer::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    ovtk_wordpiece* h = handle(inputs, 5);
    const ovtk_ragged_strings in = ragged_at(inputs);
    const int32_t unk = i32(inputs[8])[0];  // read every call (wordpiece_tokenizer.cpp:74)
    const size_t cap = inputs[4].get_size() + inputs[2].get_size();
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_ragged_i32_out out{i32(outputs[0]), i32(outputs[1]), i32(outputs[2]), int64_t(cap), 0, 0};
    check(ovtk_wordpiece_run(h, &in, unk, &out, OVTK_MEM_HOST, nullptr), "WordpieceTokenizer");
    outputs[2].set_shape(one_dim(size_t(out.n_data)));
    return true;
}

// ------------------------------------------------------------------------------------------------ VocabEncoder
VocabEncoder::VocabEncoder(const ov::OutputVector& arguments) : Base(arguments) { constructor_validate_and_infer_types(); }
void VocabEncoder::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 8, "Incorrect number of inputs passed to VocabEncoder: ", get_input_size());
    expect_strings(this, 0, "VocabEncoder");
    expect_strings(this, 3, "VocabEncoder keys");
    const Type t = get_input_element_type(6);
    OPENVINO_ASSERT(t == el::i32 || t == el::i64 || t == el::dynamic, "VocabEncoder: unsupported element type: ", t);
    set_output_type(0, t, kDyn1D);
}
std::shared_ptr<ov::Node> VocabEncoder::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<VocabEncoder>(inputs);
    c->m_state = m_state;
    return c;
}
bool VocabEncoder::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const Type t = inputs[6].get_element_type();
    OPENVINO_ASSERT(t == el::i32 || t == el::i64, "VocabEncoder: unsupported element type: ", t);
    ovtk_vocab_encoder* h = ensure(*m_state, ovtk_vocab_encoder_destroy, [&](ovtk_vocab_encoder** out) {
        const ovtk_vocab_encoder_params p{strings_at(inputs, 3), inputs[6].data(), int(t.size()), device()};
        check(ovtk_vocab_encoder_create(&p, out), "VocabEncoder (map construction)");
    });
    const ovtk_strings in = strings_at(inputs, 0);
    outputs[0].set_shape(one_dim(size_t(in.n)));  // 1-D whatever the input's shape (vocab_encoder.cpp:85)
    check(ovtk_vocab_encoder_run(h, &in, inputs[7].data(), outputs[0].data(), OVTK_MEM_HOST, nullptr), "VocabEncoder");
    return true;
}

// ------------------------------------------------------------------------------------------------ RaggedToDense
RaggedToDense::RaggedToDense(const ov::OutputVector& arguments, bool pad_right, bool pad_max_length)
    : Base(arguments), m_pad_right(pad_right), m_pad_max_length(pad_max_length) {
    constructor_validate_and_infer_types();
}
void RaggedToDense::validate_and_infer_types() {
    const size_t n = get_input_size();
    OPENVINO_ASSERT(n == 5 || n == 6, "RaggedToDense: 5 or 6 inputs expected, got ", n);
    expect_i32(this, 0, "RaggedToDense");
    expect_i32(this, 1, "RaggedToDense");
    ov::PartialShape shape = get_input_partial_shape(0);
    if (shape.rank().is_static()) {
        shape.push_back(ov::Dimension());
        const ov::PartialShape data = get_input_partial_shape(2);
        if (data.rank().is_static())
            for (int64_t k = 1; k < data.rank().get_length(); ++k) shape.push_back(data[size_t(k)]);
        else
            shape = ov::PartialShape::dynamic();
    }
    set_output_type(0, get_input_element_type(2), shape);
    set_output_type(1, el::boolean, shape);
}
std::shared_ptr<ov::Node> RaggedToDense::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<RaggedToDense>(inputs, m_pad_right, m_pad_max_length);
}
bool RaggedToDense::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("pad_right", m_pad_right);
    visitor.on_attribute("m_pad_max_length", m_pad_max_length);  // (sic: the reference's IR name, ragged_to_dense.hpp:51-55)
    return true;
}
bool RaggedToDense::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const int32_t target = i32(inputs[3])[0];  // read as i32 (ragged_to_dense.cpp:82)
    const bool pad_right = inputs.size() == 6 ? static_cast<const bool*>(inputs[5].data())[0] : m_pad_right;  // input 5 overrides (:112-115)
    ov::Shape shape = inputs[0].get_shape();
    const size_t rows = inputs[0].get_size();
    shape.push_back(size_t(target));
    const ov::Shape data_shape = inputs[2].get_shape();
    size_t inner = 1;
    for (size_t k = 1; k < data_shape.size(); ++k) {
        shape.push_back(data_shape[k]);
        inner *= data_shape[k];
    }
    outputs[0].set_shape(shape);
    outputs[1].set_shape(shape);
    check(ovtk_ragged_to_dense(i32(inputs[0]), i32(inputs[1]), int64_t(rows), inputs[2].data(), int64_t(data_shape.empty() ? 0 : data_shape[0]),
                               int(inputs[2].get_element_type().size()), int64_t(inner), target, inputs[4].data(), pad_right ? 1 : 0,
                               m_pad_max_length ? 1 : 0, outputs[0].data(), u8(outputs[1]), OVTK_MEM_HOST, device(), nullptr),
          "RaggedToDense");
    return true;
}

// ------------------------------------------------------------------------------------------------ VocabDecoder
VocabDecoder::VocabDecoder(const ov::OutputVector& arguments, std::vector<int> skip_tokens) : Base(arguments), m_skip_tokens(std::move(skip_tokens)) {
    constructor_validate_and_infer_types();
}
void VocabDecoder::validate_and_infer_types() {
    const size_t n = get_input_size();
    OPENVINO_ASSERT(n == 4 || n == 5, "Incorrect number of inputs passed to VocabDecoder: ", n);
    expect_i32(this, 0, "VocabDecoder ids");
    expect_strings(this, 1, "VocabDecoder vocab");
    const ov::PartialShape ids = get_input_partial_shape(0);
    ragged_string_outputs(this, ids.rank().is_static() && ids.rank().get_length() >= 1 ? ov::PartialShape{ids[0]} : kDyn1D);
}
std::shared_ptr<ov::Node> VocabDecoder::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<VocabDecoder>(inputs, m_skip_tokens);
    c->m_state = m_state;
    return c;
}
bool VocabDecoder::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("skip_tokens", m_skip_tokens);
    return true;
}
ovtk_vocab_decoder* VocabDecoder::handle(const ov::TensorVector& inputs) const {
    return ensure(*m_state, ovtk_vocab_decoder_destroy, [&](ovtk_vocab_decoder** out) {
        std::vector<int32_t> skip(m_skip_tokens.begin(), m_skip_tokens.end());
        const ovtk_vocab_decoder_params p{strings_at(inputs, 1), skip.data(), int64_t(skip.size()), device()};
        check(ovtk_vocab_decoder_create(&p, out), "VocabDecoder (table construction)");
    });
}
bool VocabDecoder::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    ovtk_vocab_decoder* h = handle(inputs);
    const ov::Shape ids = inputs[0].get_shape();
    OPENVINO_ASSERT(ids.size() == 2, "VocabDecoder: ids must be [batch, seq_len]");
    const size_t batch = ids[0], seq = ids[1], tokens = batch * std::max<size_t>(seq, 1);
    // chars: every token at its longest -- the vocabulary's longest token times the id count bounds the output
    const ovtk_strings vocab = strings_at(inputs, 1);
    int32_t longest = 0;
    for (int64_t k = 0; k < vocab.n; ++k) longest = std::max(longest, vocab.ends[k] - vocab.begins[k]);
    const size_t cap = std::min<size_t>(batch * seq * size_t(longest), size_t(INT32_MAX) - 1);
    outputs[0].set_shape(one_dim(batch));
    outputs[1].set_shape(one_dim(batch));
    outputs[2].set_shape(one_dim(tokens));
    outputs[3].set_shape(one_dim(tokens));
    outputs[4].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_strings_out out{i32(outputs[2]), i32(outputs[3]), u8(outputs[4]), int64_t(cap), 0};
    const bool skip_input = inputs.size() == 5;  // input 4 overrides the attribute, an empty one too (vocab_decoder.cpp:36-41)
    static const int32_t none = 0;
    const int32_t* skip = skip_input ? (inputs[4].get_size() ? i32(inputs[4]) : &none) : nullptr;
    check(ovtk_vocab_decoder_run(h, i32(inputs[0]), int64_t(batch), int64_t(seq), skip, skip_input ? int64_t(inputs[4].get_size()) : 0,
                                 i32(outputs[0]), i32(outputs[1]), &out, OVTK_MEM_HOST, nullptr),
          "VocabDecoder");
    outputs[4].set_shape(one_dim(size_t(out.n_chars)));
    return true;
}

// ------------------------------------------------------------------------------------------------ ByteFallback / FuzeRagged
void ByteFallback::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 3, "ByteFallback: 3 inputs expected");
    expect_strings(this, 0, "ByteFallback");
    string_outputs(this, 0, get_input_partial_shape(0));
}
bool ByteFallback::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const ovtk_strings in = strings_at(inputs, 0);
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(size_t(in.n_chars), 1)));  // byte_fallback.cpp:24
    ovtk_strings_out out{i32(outputs[0]), i32(outputs[1]), u8(outputs[2]), in.n_chars, 0};
    check(ovtk_byte_fallback(&in, &out, OVTK_MEM_HOST, device(), nullptr), "ByteFallback");
    outputs[2].set_shape(one_dim(size_t(out.n_chars)));
    return true;
}
void FuzeRagged::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 4, "FuzeRagged: 4 inputs expected");
    for (size_t k = 0; k < 4; ++k) expect_i32(this, k, "FuzeRagged");
    set_output_type(0, el::i32, get_input_partial_shape(0));
    set_output_type(1, el::i32, get_input_partial_shape(0));
}
bool FuzeRagged::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    check(ovtk_fuze_ragged(i32(inputs[0]), i32(inputs[1]), int64_t(inputs[0].get_size()), i32(inputs[2]), i32(inputs[3]),
                           int64_t(inputs[2].get_size()), i32(outputs[0]), i32(outputs[1]), OVTK_MEM_HOST, device(), nullptr),
          "FuzeRagged");
    return true;
}

// ------------------------------------------------------------------------------------------------ Truncate / CombineSegments
Truncate::Truncate(const ov::OutputVector& arguments) : Base(arguments) { constructor_validate_and_infer_types(); }
void Truncate::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() > 3, "Truncate: ragged tensor(s) + max_length, side, [mode] expected");
    OPENVINO_ASSERT(get_input_size() == 6 || get_input_size() == 9, "Truncate: one or two ragged tensors + max_length, side, mode expected");
    m_num_inputs = int64_t(get_input_size() - 3) / 3;  // (begins, ends, data) x 1 or 2, then max_length, side, mode
    for (int64_t k = 0; k < m_num_inputs; ++k) {
        set_output_type(size_t(3 * k), el::i32, get_input_partial_shape(size_t(3 * k)));
        set_output_type(size_t(3 * k + 1), el::i32, get_input_partial_shape(size_t(3 * k)));
        set_output_type(size_t(3 * k + 2), get_input_element_type(size_t(3 * k + 2)), get_input_partial_shape(size_t(3 * k + 2)));
    }
}
std::shared_ptr<ov::Node> Truncate::clone_with_new_inputs(const ov::OutputVector& inputs) const { return std::make_shared<Truncate>(inputs); }
bool Truncate::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("m_num_inputs", m_num_inputs);
    return true;
}
bool Truncate::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const size_t n = inputs.size();
    const int32_t max_length = i32(inputs[n - 3])[0];
    const std::string side = text_of(inputs[n - 2]), mode = text_of(inputs[n - 1]);
    const int k = int(m_num_inputs);
    OPENVINO_ASSERT(k == 1 || k == 2, "Only single or pair inputs are supported in Truncation op");
    for (int q = 0; q < k; ++q) {  // the element tensors pass through; begins / ends are rewritten (truncate.cpp:48-52)
        outputs[size_t(3 * q)].set_shape(inputs[size_t(3 * q)].get_shape());
        outputs[size_t(3 * q + 1)].set_shape(inputs[size_t(3 * q)].get_shape());
        outputs[size_t(3 * q + 2)] = inputs[size_t(3 * q + 2)];
    }
    check(ovtk_truncate(k, i32(inputs[0]), i32(inputs[1]), k == 2 ? i32(inputs[3]) : nullptr, k == 2 ? i32(inputs[4]) : nullptr,
                        int64_t(inputs[0].get_size()), max_length, side.c_str(), mode.c_str(), i32(outputs[0]), i32(outputs[1]),
                        k == 2 ? i32(outputs[3]) : nullptr, k == 2 ? i32(outputs[4]) : nullptr, OVTK_MEM_HOST, device(), nullptr),
          "Truncate");
    return true;
}
void CombineSegments::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() > 0 && (get_input_size() - 1) % 3 == 0, "CombineSegments: k ragged tensors + segment ids expected");
    ov::PartialShape rows = kDyn1D;
    for (size_t k = 0; k + 1 < get_input_size(); k += 3)
        if (get_input_partial_shape(k).rank().is_static() && get_input_partial_shape(k).rank().get_length() > 0) rows = get_input_partial_shape(k);
    ragged_i32_outputs(this, rows);
    set_output_type(2, get_input_element_type(2), kDyn1D);
    set_output_type(3, el::i32, rows);
    set_output_type(4, el::i32, rows);
    set_output_type(5, get_input_element_type(get_input_size() - 1), kDyn1D);
}
bool CombineSegments::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const size_t k = (inputs.size() - 1) / 3;
    OPENVINO_ASSERT(inputs[2].get_element_type() == el::i32, "CombineSegments: this library combines i32 segments (token ids)");
    std::vector<ovtk_ragged_i32> segs(k);
    size_t rows = 1, total = 0;
    for (size_t q = 0; q < k; ++q) {
        segs[q] = ovtk_ragged_i32{i32(inputs[3 * q]), i32(inputs[3 * q + 1]), i32(inputs[3 * q + 2]), int64_t(inputs[3 * q].get_size()),
                                  int64_t(inputs[3 * q + 2].get_size())};
        rows = std::max(rows, inputs[3 * q].get_size());
    }
    for (size_t q = 0; q < k; ++q) total += inputs[3 * q + 2].get_size() * (inputs[3 * q].get_size() == 1 ? rows : 1);
    for (size_t o : {size_t(0), size_t(1), size_t(3), size_t(4)}) outputs[o].set_shape(one_dim(rows));
    outputs[2].set_shape(one_dim(std::max<size_t>(total, 1)));
    outputs[5].set_shape(one_dim(std::max<size_t>(total, 1)));
    int64_t n_out = 0;
    check(ovtk_combine_segments(segs.data(), int(k), i32(inputs.back()), i32(outputs[0]), i32(outputs[1]), i32(outputs[2]), i32(outputs[5]),
                                int64_t(total), &n_out, OVTK_MEM_HOST, device(), nullptr),
          "CombineSegments");
    std::memcpy(outputs[3].data(), outputs[0].data(), rows * 4);  // the segment-id tensor is ragged over the same rows (:33)
    std::memcpy(outputs[4].data(), outputs[1].data(), rows * 4);
    outputs[2].set_shape(one_dim(size_t(n_out)));
    outputs[5].set_shape(one_dim(size_t(n_out)));
    return true;
}

// ------------------------------------------------------------------------------------------------ UTF8Validate / TrieTokenizer
UTF8Validate::UTF8Validate(const ov::OutputVector& arguments, bool replace_mode) : Base(arguments), m_replace_mode(replace_mode) {
    constructor_validate_and_infer_types();
}
void UTF8Validate::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 3, "UTF8Validate: 3 inputs expected");
    expect_strings(this, 0, "UTF8Validate");
    string_outputs(this, 0, get_input_partial_shape(0));
}
std::shared_ptr<ov::Node> UTF8Validate::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<UTF8Validate>(inputs, m_replace_mode);
}
bool UTF8Validate::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("replace_mode", m_replace_mode);
    return true;
}
bool UTF8Validate::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    const ovtk_strings in = strings_at(inputs, 0);
    const size_t cap = size_t(in.n_chars) * 3;  // one byte becomes at most U+FFFD's three (utf8_validate.cpp:31-33)
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_strings_out out{i32(outputs[0]), i32(outputs[1]), u8(outputs[2]), int64_t(cap), 0};
    check(ovtk_utf8_validate(&in, m_replace_mode ? 1 : 0, &out, OVTK_MEM_HOST, device(), nullptr), "UTF8Validate");
    outputs[2].set_shape(one_dim(size_t(out.n_chars)));
    return true;
}
TrieTokenizer::TrieTokenizer(const ov::OutputVector& arguments) : Base(arguments) { constructor_validate_and_infer_types(); }
void TrieTokenizer::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 9, "TrieTokenizer: 9 inputs expected");
    expect_ragged_strings(this, "TrieTokenizer");
    expect_strings(this, 5, "TrieTokenizer vocab");
    expect_i32(this, 8, "TrieTokenizer indices");
    ragged_i32_outputs(this, get_input_partial_shape(0));
}
std::shared_ptr<ov::Node> TrieTokenizer::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    auto c = std::make_shared<TrieTokenizer>(inputs);
    c->m_state = m_state;
    return c;
}
bool TrieTokenizer::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    ovtk_trie_tokenizer* h = ensure(*m_state, ovtk_trie_tokenizer_destroy, [&](ovtk_trie_tokenizer** out) {
        OPENVINO_ASSERT(inputs[5].get_size() == inputs[8].get_size(), "Vocab size must be equal to Indices size");
        const ovtk_strings vocab = strings_at(inputs, 5);
        check(ovtk_trie_tokenizer_create(&vocab, i32(inputs[8]), device(), out), "TrieTokenizer (trie construction)");
    });
    const ovtk_ragged_strings in = ragged_at(inputs);
    const size_t cap = inputs[4].get_size();  // trie_tokenizer.cpp:60
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_ragged_i32_out out{i32(outputs[0]), i32(outputs[1]), i32(outputs[2]), int64_t(cap), 0, 0};
    check(ovtk_trie_tokenizer_run(h, &in, &out, OVTK_MEM_HOST, nullptr), "TrieTokenizer");
    outputs[2].set_shape(one_dim(size_t(out.n_data)));
    return true;
}

// ------------------------------------------------------------------------------------------------ StringTensorUnpack (u8 wire form)
StringTensorUnpack::StringTensorUnpack(const ov::OutputVector& arguments, const std::string& mode) : Base(arguments), m_mode(mode) {
    constructor_validate_and_infer_types();
}
void StringTensorUnpack::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 1, "Number of inputs for StringTensorUnpack is not equal to 1");
    OPENVINO_ASSERT(m_mode == "begins_ends", "StringTensorUnpack supports only 'begins_ends' mode, but get ", m_mode);
    OPENVINO_ASSERT(get_input_element_type(0) == el::u8 || get_input_element_type(0) == el::dynamic,
                    "StringTensorUnpack: this library takes the packed u8 form; element::string tensors stay with the stock op");
    string_outputs(this, 0, kDyn1D);
}
std::shared_ptr<ov::Node> StringTensorUnpack::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<StringTensorUnpack>(inputs, m_mode);
}
bool StringTensorUnpack::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("mode", m_mode);
    return true;
}
bool StringTensorUnpack::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    // The wire is host memory and so are this op's outputs under the CPU plugin: the header already holds the offsets, so
    // the decomposed tensors are views of it (parse_packed_strings, utils.cpp:18-29) -- nothing for a GPU to compute.  The
    // device-side form of this step (one buffer over PCIe, begins / ends / chars only ever in HBM) is
    // ovtk_string_tensor_unpack, used by callers that keep the batch on the device.
    const size_t bytes = inputs[0].get_byte_size();
    OPENVINO_ASSERT(bytes >= 4, "Incorrect packed string tensor format: no batch size in the packed string tensor");
    const uint8_t* p = u8(inputs[0]);
    int32_t n = 0;
    std::memcpy(&n, p, 4);
    OPENVINO_ASSERT(n >= 0 && bytes >= 4 + 4 + 4 * size_t(n), "Incorrect packed string tensor format: the packed string tensor must contain first "
                                                               "string offset and end indices");
    const int32_t* offsets = reinterpret_cast<const int32_t*>(p + 4);
    const size_t n_chars = bytes - (8 + 4 * size_t(n));
    outputs[0].set_shape(one_dim(size_t(n)));
    outputs[1].set_shape(one_dim(size_t(n)));
    outputs[2].set_shape(one_dim(n_chars));
    if (n) {
        std::memcpy(outputs[0].data(), offsets, 4 * size_t(n));      // begins = [begin_0, end_0 .. end_{n-2}]
        std::memcpy(outputs[1].data(), offsets + 1, 4 * size_t(n));  // ends
    }
    if (n_chars) std::memcpy(outputs[2].data(), p + 8 + 4 * size_t(n), n_chars);
    return true;
}

// ------------------------------------------------------------------------------------------------ StringTensorPack
StringTensorPack::StringTensorPack(const ov::OutputVector& arguments, const std::string& mode)
    : ov::op::v15::StringTensorPack(arguments.at(0), arguments.at(1), arguments.at(2)), m_mode(mode) {
    constructor_validate_and_infer_types();
}
void StringTensorPack::validate_and_infer_types() {
    OPENVINO_ASSERT(m_mode == "begins_ends", "StringTensorPack supports only 'begins_ends' mode, but get ", m_mode);
    ov::op::v15::StringTensorPack::validate_and_infer_types();
}
std::shared_ptr<ov::Node> StringTensorPack::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<StringTensorPack>(inputs, m_mode);
}
bool StringTensorPack::visit_attributes(ov::AttributeVisitor& visitor) {
    visitor.on_attribute("mode", m_mode);
    return true;
}

// ------------------------------------------------------------------------------------------------ the fused nodes (fuse_pass.cpp)
FusedSplitBPE::FusedSplitBPE(const ov::OutputVector& arguments, std::shared_ptr<const SpecialTokensSplit> special, std::shared_ptr<const RegexSplit> split,
                             std::shared_ptr<const BPETokenizer> bpe, bool has_skips, size_t bpe_inputs)
    : Base(arguments), m_special(std::move(special)), m_split(std::move(split)), m_bpe(std::move(bpe)), m_has_skips(has_skips), m_bpe_inputs(bpe_inputs) {
    constructor_validate_and_infer_types();
}
void FusedSplitBPE::validate_and_infer_types() {
    expect_ragged_strings(this, "OvtkFusedSplitBPE");
    ragged_i32_outputs(this, get_input_partial_shape(0));
}
std::shared_ptr<ov::Node> FusedSplitBPE::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<FusedSplitBPE>(inputs, m_special, m_split, m_bpe, m_has_skips, m_bpe_inputs);
}
bool FusedSplitBPE::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    size_t at = 5 + (m_has_skips ? 1 : 0);
    ovtk_special_tokens_split* special = m_special ? m_special->handle(inputs[at++]) : nullptr;
    ovtk_regex_split* split = m_split->handle(inputs[at++]);
    ovtk_bpe* bpe = m_bpe->handle(inputs, at, m_bpe_inputs);
    const ovtk_ragged_strings in = ragged_at(inputs);
    const size_t cap = (inputs[4].get_size() + inputs[2].get_size()) * (1 + m_bpe->end_suffix_size());   // bpe_tokenizer.cpp:135
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_ragged_i32_out out{i32(outputs[0]), i32(outputs[1]), i32(outputs[2]), int64_t(cap), 0, 0};
    const uint8_t* skips = m_has_skips ? u8(inputs[5]) : nullptr;
    if (special) check(ovtk_encode_special_run(special, split, bpe, &in, skips, &out, OVTK_MEM_HOST, nullptr), "OvtkFusedSplitBPE");
    else check(ovtk_encode_run(split, bpe, &in, skips, &out, OVTK_MEM_HOST, nullptr), "OvtkFusedSplitBPE");
    outputs[0].set_shape(one_dim(size_t(out.n_rows)));
    outputs[1].set_shape(one_dim(size_t(out.n_rows)));
    outputs[2].set_shape(one_dim(size_t(out.n_data)));
    return true;
}

FusedSplitWordpiece::FusedSplitWordpiece(const ov::OutputVector& arguments, std::shared_ptr<const RegexSplit> whitespace,
                                         std::shared_ptr<const RegexSplit> delimiters, std::shared_ptr<const WordpieceTokenizer> wordpiece)
    : Base(arguments), m_whitespace(std::move(whitespace)), m_delimiters(std::move(delimiters)), m_wordpiece(std::move(wordpiece)) {
    constructor_validate_and_infer_types();
}
void FusedSplitWordpiece::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 11, "OvtkFusedSplitWordpiece: ragged strings (5), two patterns, vocabulary (3), unk_token_id expected");
    expect_ragged_strings(this, "OvtkFusedSplitWordpiece");
    ragged_i32_outputs(this, get_input_partial_shape(0));
}
std::shared_ptr<ov::Node> FusedSplitWordpiece::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<FusedSplitWordpiece>(inputs, m_whitespace, m_delimiters, m_wordpiece);
}
bool FusedSplitWordpiece::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    ovtk_regex_split* ws = m_whitespace->handle(inputs[5]);
    ovtk_regex_split* pu = m_delimiters->handle(inputs[6]);
    ovtk_wordpiece* wp = m_wordpiece->handle(inputs, 7);
    const ovtk_ragged_strings in = ragged_at(inputs);
    const int32_t unk = i32(inputs[10])[0];   // read every call (wordpiece_tokenizer.cpp:74)
    const size_t cap = inputs[4].get_size() + inputs[2].get_size();
    outputs[0].set_shape(inputs[0].get_shape());
    outputs[1].set_shape(inputs[0].get_shape());
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_ragged_i32_out out{i32(outputs[0]), i32(outputs[1]), i32(outputs[2]), int64_t(cap), 0, 0};
    check(ovtk_wordpiece_encode_run(wp, ws, pu, &in, unk, &out, OVTK_MEM_HOST, nullptr), "OvtkFusedSplitWordpiece");
    outputs[0].set_shape(one_dim(size_t(out.n_rows)));
    outputs[1].set_shape(one_dim(size_t(out.n_rows)));
    outputs[2].set_shape(one_dim(size_t(out.n_data)));
    return true;
}

FusedDetokenize::FusedDetokenize(const ov::OutputVector& arguments, std::shared_ptr<const VocabDecoder> decoder, bool byte_fallback)
    : Base(arguments), m_decoder(std::move(decoder)), m_byte_fallback(byte_fallback) {
    constructor_validate_and_infer_types();
}
void FusedDetokenize::validate_and_infer_types() {
    OPENVINO_ASSERT(get_input_size() == 4 || get_input_size() == 5, "OvtkFusedDetokenize: VocabDecoder's inputs expected");
    string_outputs(this, 0, ov::PartialShape{ov::Dimension()});
}
std::shared_ptr<ov::Node> FusedDetokenize::clone_with_new_inputs(const ov::OutputVector& inputs) const {
    return std::make_shared<FusedDetokenize>(inputs, m_decoder, m_byte_fallback);
}
bool FusedDetokenize::evaluate(ov::TensorVector& outputs, const ov::TensorVector& inputs) const {
    ovtk_vocab_decoder* h = m_decoder->handle(inputs);
    const ov::Shape ids = inputs[0].get_shape();
    OPENVINO_ASSERT(ids.size() == 2, "OvtkFusedDetokenize: ids must be [batch, seq_len]");
    const size_t batch = ids[0], seq = ids[1];
    const ovtk_strings vocab = strings_at(inputs, 1);
    int32_t longest = 0;
    for (int64_t k = 0; k < vocab.n; ++k) longest = std::max(longest, vocab.ends[k] - vocab.begins[k]);
    const size_t cap = std::min<size_t>(batch * seq * size_t(longest), size_t(INT32_MAX) - 1);
    outputs[0].set_shape(one_dim(batch));
    outputs[1].set_shape(one_dim(batch));
    outputs[2].set_shape(one_dim(std::max<size_t>(cap, 1)));
    ovtk_strings_out out{i32(outputs[0]), i32(outputs[1]), u8(outputs[2]), int64_t(cap), 0};
    const bool skip_input = inputs.size() == 5;   // input 4 overrides the attribute, an empty one too (vocab_decoder.cpp:36-41)
    static const int32_t none = 0;
    const int32_t* skip = skip_input ? (inputs[4].get_size() ? i32(inputs[4]) : &none) : nullptr;
    check(ovtk_detokenize_run(h, i32(inputs[0]), int64_t(batch), int64_t(seq), skip, skip_input ? int64_t(inputs[4].get_size()) : 0,
                              m_byte_fallback ? 1 : 0, &out, OVTK_MEM_HOST, nullptr),
          "OvtkFusedDetokenize");
    outputs[2].set_shape(one_dim(size_t(out.n_chars)));
    return true;
}

}  // namespace ovtk_adapter

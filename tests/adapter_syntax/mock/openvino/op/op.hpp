// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <cstdint>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "openvino/core/except.hpp"
#include "openvino/runtime/tensor.hpp"
namespace ov {
class Node;
template <class T>
class Output;
template <class T>
class Input;
template <>
class Output<Node> {
public:
    Output() = default;
    Output(const std::shared_ptr<Node>&) {}
    std::shared_ptr<Node> get_node_shared_ptr() const;
    size_t get_index() const;
    std::set<Input<Node>> get_target_inputs() const;
    void replace(const Output<Node>& replacement);
    const element::Type& get_element_type() const;
    const PartialShape& get_partial_shape() const;
};
using OutputVector = std::vector<Output<Node>>;
class AttributeVisitor {
public:
    template <class T>
    void on_attribute(const std::string& name, T& value);
};
class DiscreteTypeInfo {
public:
    const char* name;
    const char* version_id;
    const DiscreteTypeInfo* parent;
};
class Node : public std::enable_shared_from_this<Node> {
public:
    Node() = default;
    explicit Node(const OutputVector&) {}
    virtual ~Node() = default;
    virtual void validate_and_infer_types() {}
    virtual std::shared_ptr<Node> clone_with_new_inputs(const OutputVector& inputs) const = 0;
    virtual bool visit_attributes(AttributeVisitor&) { return false; }
    virtual bool evaluate(TensorVector& outputs, const TensorVector& inputs) const { return false; }
    virtual bool has_evaluate() const { return false; }
    void constructor_validate_and_infer_types();
    size_t get_input_size() const;
    const element::Type& get_input_element_type(size_t i) const;
    const PartialShape& get_input_partial_shape(size_t i) const;
    void set_output_type(size_t i, const element::Type& type, const PartialShape& shape);
    void set_output_size(size_t n);
    OutputVector outputs();
    size_t get_output_size() const;
    Output<Node> input_value(size_t i) const;
    Output<Node> output(size_t i);
    const std::string& get_friendly_name() const;
    void set_friendly_name(const std::string& name);
};
namespace op {
class Op : public Node {
public:
    Op() = default;
    explicit Op(const OutputVector& arguments) : Node(arguments) {}
};
}  // namespace op
}  // namespace ov
#define OPENVINO_OP_MOCK_1(name) \
    static const ::ov::DiscreteTypeInfo& get_type_info_static() { static const ::ov::DiscreteTypeInfo t{name, "extension", nullptr}; return t; }
#define OPENVINO_OP_MOCK_2(name, ver) OPENVINO_OP_MOCK_1(name)
#define OPENVINO_OP_MOCK_3(name, ver, parent) OPENVINO_OP_MOCK_1(name)
#define OPENVINO_OP_MOCK_PICK(_1, _2, _3, which, ...) which
#define OPENVINO_OP(...) OPENVINO_OP_MOCK_PICK(__VA_ARGS__, OPENVINO_OP_MOCK_3, OPENVINO_OP_MOCK_2, OPENVINO_OP_MOCK_1)(__VA_ARGS__)

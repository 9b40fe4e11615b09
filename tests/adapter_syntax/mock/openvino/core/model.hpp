// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <memory>
#include <set>
#include <vector>

#include "openvino/op/op.hpp"
namespace ov {
class Model {
public:
    std::vector<std::shared_ptr<Node>> get_ordered_ops() const;
};
template <class T>
class Input;
template <>
class Input<Node> {
public:
    Node* get_node() const;
    size_t get_index() const;
    void replace_source_output(const Output<Node>& new_source) const;
};
template <class T>
std::shared_ptr<T> as_type_ptr(const std::shared_ptr<Node>& n);
void copy_runtime_info(const std::shared_ptr<Node>& from, const std::shared_ptr<Node>& to);
}  // namespace ov

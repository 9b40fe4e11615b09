// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include "openvino/op/op.hpp"
namespace ov {
namespace op {
namespace v15 {
class StringTensorPack : public Op {
public:
    OPENVINO_OP("StringTensorPack", "opset15");
    StringTensorPack() = default;
    StringTensorPack(const Output<Node>& begins, const Output<Node>& ends, const Output<Node>& symbols);
    void validate_and_infer_types() override;
    std::shared_ptr<Node> clone_with_new_inputs(const OutputVector& inputs) const override;
    bool evaluate(TensorVector& outputs, const TensorVector& inputs) const override;
    bool has_evaluate() const override;
};
}  // namespace v15
}  // namespace op
}  // namespace ov

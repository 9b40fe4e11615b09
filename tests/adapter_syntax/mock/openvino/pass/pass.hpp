// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <memory>

#include "openvino/core/model.hpp"
namespace ov {
namespace pass {
class PassBase {
public:
    virtual ~PassBase() = default;
};
class ModelPass : public PassBase {
public:
    virtual bool run_on_model(const std::shared_ptr<ov::Model>& m) = 0;
};
}  // namespace pass
}  // namespace ov
#define OPENVINO_MODEL_PASS_RTTI(name) static const char* get_type_info_static_name() { return name; }

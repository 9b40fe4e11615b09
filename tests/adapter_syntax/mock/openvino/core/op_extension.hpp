// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include "openvino/core/extension.hpp"
namespace ov {
template <class T>
class OpExtension : public Extension {
public:
    OpExtension() { (void)T::get_type_info_static(); }
};
}  // namespace ov

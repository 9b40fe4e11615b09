// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <memory>
#include <vector>
namespace ov {
class Extension {
public:
    using Ptr = std::shared_ptr<Extension>;
    virtual ~Extension() = default;
};
}  // namespace ov
#define OPENVINO_EXTENSION_C_API extern "C" __attribute__((visibility("default")))
#define OPENVINO_API_C(...) extern "C" __attribute__((visibility("default"))) __VA_ARGS__
#define OPENVINO_CREATE_EXTENSIONS(extensions)                                           \
    OPENVINO_EXTENSION_C_API void create_extensions(std::vector<::ov::Extension::Ptr>& ext); \
    OPENVINO_EXTENSION_C_API void create_extensions(std::vector<::ov::Extension::Ptr>& ext) { ext = extensions; }

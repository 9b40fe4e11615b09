// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <map>
#include <string>
namespace ov {
class Any {
public:
    Any() = default;
    template <class T>
    Any(const T&) {}
    template <class T>
    bool is() const;
    template <class T>
    T& as() const;
};
using AnyMap = std::map<std::string, Any>;
}  // namespace ov

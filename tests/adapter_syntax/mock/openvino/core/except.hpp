// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
namespace ov {
class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string& what) : std::runtime_error(what) {}
};
namespace mock_detail {
template <class... Args>
std::string concat(Args&&... args) {
    std::ostringstream s;
    (void)std::initializer_list<int>{((s << args), 0)...};
    return s.str();
}
}  // namespace mock_detail
}  // namespace ov
#define OPENVINO_THROW(...) throw ::ov::Exception(::ov::mock_detail::concat(__VA_ARGS__))
#define OPENVINO_ASSERT(cond, ...)                                            \
    do {                                                                      \
        if (!(cond)) throw ::ov::Exception(::ov::mock_detail::concat(#cond, ": ", ##__VA_ARGS__)); \
    } while (0)

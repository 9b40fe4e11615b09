// SYNTAX-ONLY MOCK (tests/adapter_syntax/README.md): declarations shaped like OpenVINO's public API, no behaviour.
#pragma once
#include <cstddef>
#include <cstdint>
#include <initializer_list>
#include <memory>
#include <ostream>
#include <string>
#include <vector>
namespace ov {
class Shape : public std::vector<size_t> {
public:
    using std::vector<size_t>::vector;
};
class Dimension {
public:
    Dimension() = default;
    Dimension(int64_t) {}
    bool is_dynamic() const;
    bool is_static() const;
    int64_t get_length() const;
};
class Rank : public Dimension {
public:
    using Dimension::Dimension;
};
class PartialShape {
public:
    PartialShape() = default;
    PartialShape(std::initializer_list<Dimension>) {}
    PartialShape(const Shape&) {}
    static PartialShape dynamic(Rank r = Rank());
    Rank rank() const;
    void push_back(const Dimension&);
    const Dimension& operator[](size_t) const;
    Dimension& operator[](size_t);
    std::string to_string() const;
};
namespace element {
enum class Type_t { dynamic, boolean, i32, i64, u8, string };
class Type {
public:
    constexpr Type() = default;
    constexpr Type(Type_t t) : m_t(t) {}
    size_t size() const;
    std::string get_type_name() const;
    bool operator==(const Type& o) const { return m_t == o.m_t; }
    bool operator!=(const Type& o) const { return m_t != o.m_t; }
private:
    Type_t m_t = Type_t::dynamic;
};
std::ostream& operator<<(std::ostream&, const Type&);
constexpr Type dynamic(Type_t::dynamic), boolean(Type_t::boolean), i32(Type_t::i32), i64(Type_t::i64), u8(Type_t::u8), string(Type_t::string);
}  // namespace element
class Tensor {
public:
    Tensor() = default;
    Tensor(const element::Type&, const Shape&);
    void* data(const element::Type& = {}) const;
    template <class T>
    T* data() const;
    size_t get_size() const;
    size_t get_byte_size() const;
    const Shape& get_shape() const;
    void set_shape(const Shape&);
    const element::Type& get_element_type() const;
    explicit operator bool() const noexcept;
};
using TensorVector = std::vector<Tensor>;
}  // namespace ov

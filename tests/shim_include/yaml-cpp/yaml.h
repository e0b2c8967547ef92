// TEST-ONLY compile shim (see ../README.md): the slice of yaml-cpp's Node the reference's initConfig
// (registration.cpp:212-230) and the adapter's fromYaml use -- operator[] by key, as<T>(), scalar assignment.
#pragma once
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
namespace YAML {
class Node {
 public:
  Node() : d_(std::make_shared<Data>()) {}
  Node operator[](const std::string& key) {
    auto& slot = d_->map[key];
    if (!slot) slot = std::make_shared<Data>();
    return Node(slot);
  }
  const Node operator[](const std::string& key) const {
    auto it = d_->map.find(key);
    if (it == d_->map.end()) throw std::runtime_error("yaml shim: no key " + key);
    return Node(it->second);
  }
  template <class T>
  Node& operator=(const T& v) {
    std::ostringstream os;
    os.precision(17);
    os << v;
    d_->scalar = os.str();
    return *this;
  }
  template <class T>
  T as() const {
    std::istringstream is(d_->scalar);
    T v{};
    if (!(is >> v)) throw std::runtime_error("yaml shim: bad conversion of '" + d_->scalar + "'");
    return v;
  }

 private:
  struct Data {
    std::string scalar;
    std::map<std::string, std::shared_ptr<Data>> map;
  };
  explicit Node(std::shared_ptr<Data> d) : d_(std::move(d)) {}
  std::shared_ptr<Data> d_;
};
template <>
inline std::string Node::as<std::string>() const { return d_->scalar; }
}  // namespace YAML

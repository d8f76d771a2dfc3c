// The torchbind class `_core_C.ScalarType` for a process in which the reference's own `_core_C` extension is absent
// (kernels/core/torch_bindings.cpp:10-13 registers it there; kernels/core/scalar_type.hpp describes the type).  The custom-op
// schemas of the quantised GEMMs name this class (`__torch__.torch.classes._core_C.ScalarType b_q_type`,
// kernels/torch_bindings.cpp:195-201); with it registered the MI355X library can expose the verbatim schema standalone.
// Loaded by torch_cpp.load() ONLY when `torch.classes._core_C.ScalarType` does not resolve -- a class can be registered once.
//
// A sub-byte / quantised scalar type: `exponent` and `mantissa` field widths (integers: exponent 0, mantissa = the value
// bits without the sign), a sign flag, an integer `bias` (stored value = real value + bias: uint4b8 stores x + 8), and for
// floats whether infinities exist and how NaNs are encoded.  Same Python surface as the reference's class: the properties
// mantissa / exponent / bias / signed / size_bits, the predicates, min() / max(), __str__ / __repr__ / __eq__, the
// convenience constructors int_ / uint / float_IEEE754 / float_, and the __obj_flatten__ pair used by torch.compile.
#include <torch/custom_class.h>
#include <torch/library.h>

#include <cmath>
#include <string>
#include <tuple>

namespace {

enum NanKind : int64_t { kNanNone = 0, kNanIeee = 1, kNanExtendedRange = 2 };

struct QuantScalarType : torch::CustomClassHolder {
  int64_t exponent_, mantissa_, bias_;
  bool signed_, finite_only_;
  int64_t nan_kind_;

  QuantScalarType(int64_t exponent, int64_t mantissa, int64_t bias, bool is_signed, bool finite_only = false,
                  int64_t nan_kind = kNanIeee)
      : exponent_(exponent), mantissa_(mantissa), bias_(bias), signed_(is_signed), finite_only_(finite_only), nan_kind_(nan_kind) {
    TORCH_CHECK(exponent >= 0 && exponent < 256 && mantissa >= 0 && mantissa < 256, "ScalarType: field widths must fit 8 bits");
    TORCH_CHECK(nan_kind >= kNanNone && nan_kind <= kNanExtendedRange, "ScalarType: invalid NaN representation ", nan_kind);
  }

  using Ptr = c10::intrusive_ptr<QuantScalarType>;
  static Ptr make(int64_t e, int64_t m, int64_t b, bool s, bool f = false, int64_t n = kNanIeee) {
    return c10::make_intrusive<QuantScalarType>(e, m, b, s, f, n);
  }
  static Ptr make_int(int64_t size_bits, c10::optional<int64_t> bias) {
    TORCH_CHECK(size_bits >= 2, "ScalarType.int_: at least a sign and one value bit");
    return make(0, size_bits - 1, bias.value_or(0), true);
  }
  static Ptr make_uint(int64_t size_bits, c10::optional<int64_t> bias) {
    TORCH_CHECK(size_bits >= 1, "ScalarType.uint: at least one bit");
    return make(0, size_bits, bias.value_or(0), false);
  }
  static Ptr make_float_ieee(int64_t exponent, int64_t mantissa) {
    TORCH_CHECK(exponent > 0 && mantissa > 0, "ScalarType.float_IEEE754: exponent and mantissa widths must be positive");
    return make(exponent, mantissa, 0, true, false, kNanIeee);
  }
  static Ptr make_float(int64_t exponent, int64_t mantissa, bool finite_only, int64_t nan_kind) {
    TORCH_CHECK(exponent > 0 && mantissa > 0, "ScalarType.float_: exponent and mantissa widths must be positive");
    TORCH_CHECK(nan_kind != kNanIeee, "ScalarType.float_: use float_IEEE754 for types that follow IEEE 754");
    return make(exponent, mantissa, 0, true, finite_only, nan_kind);
  }

  int64_t size_bits() const { return exponent_ + mantissa_ + (signed_ ? 1 : 0); }
  bool is_signed() const { return signed_; }
  bool is_integer() const { return exponent_ == 0; }
  bool is_floating_point() const { return exponent_ > 0; }
  bool is_ieee_754() const { return is_floating_point() && !finite_only_ && nan_kind_ == kNanIeee; }
  bool has_nans() const { return is_floating_point() && nan_kind_ != kNanNone; }
  bool has_infs() const { return is_floating_point() && !finite_only_; }
  bool has_bias() const { return bias_ != 0; }

  // largest finite magnitude of a float type: all-ones exponent is reserved unless the type is finite-only, in which case the
  // top exponent is a normal binade -- minus its all-ones mantissa when that code is the NaN (the e4m3fn convention)
  double float_max() const {
    const int64_t ebias = (int64_t(1) << (exponent_ - 1)) - 1;
    int64_t top = (int64_t(1) << exponent_) - 1;
    double frac = 2.0 - std::ldexp(1.0, -(int)mantissa_);
    if (!finite_only_) top -= 1;                                         // infinities (and IEEE NaNs) own the last binade
    else if (nan_kind_ == kNanExtendedRange) frac -= std::ldexp(1.0, -(int)mantissa_);
    return std::ldexp(frac, (int)(top - ebias));
  }
  c10::IValue max_value() const {
    if (is_floating_point()) return c10::IValue(float_max());
    return c10::IValue(((int64_t(1) << mantissa_) - 1) - bias_);
  }
  c10::IValue min_value() const {
    if (is_floating_point()) return c10::IValue(-float_max());
    return c10::IValue((signed_ ? -(int64_t(1) << mantissa_) : int64_t(0)) - bias_);
  }

  std::string name() const {
    if (is_floating_point()) {
      std::string s = "float" + std::to_string(size_bits()) + "_e" + std::to_string(exponent_) + "m" + std::to_string(mantissa_);
      if (!is_ieee_754()) {
        if (finite_only_) s += "f";
        if (nan_kind_ != kNanNone) s += "n";
      }
      return s;
    }
    std::string s = (signed_ ? "int" : "uint") + std::to_string(size_bits());
    if (has_bias()) s += "b" + std::to_string(bias_);
    return s;
  }
  bool same(const QuantScalarType& o) const {
    return exponent_ == o.exponent_ && mantissa_ == o.mantissa_ && bias_ == o.bias_ && signed_ == o.signed_ &&
           finite_only_ == o.finite_only_ && nan_kind_ == o.nan_kind_;
  }

  // one integer that holds every field (torch.compile flattens the object to it and rebuilds it)
  int64_t id() const {
    return exponent_ | (mantissa_ << 8) | (int64_t(signed_) << 16) | (int64_t(finite_only_) << 17) | (nan_kind_ << 18) |
           ((bias_ & 0xffffffffLL) << 24);
  }
  static Ptr from_id(int64_t v) {
    const int64_t b = (int64_t)(int32_t)((v >> 24) & 0xffffffffLL);
    return make(v & 0xff, (v >> 8) & 0xff, b, (v >> 16) & 1, (v >> 17) & 1, (v >> 18) & 3);
  }
};

}  // namespace

TORCH_LIBRARY(_core_C, lib) {
  using S = QuantScalarType;
  using P = c10::intrusive_ptr<S>;
  lib.class_<S>("ScalarType")
      .def(torch::init([](int64_t exponent, int64_t mantissa, int64_t bias, bool is_signed) {
        return S::make(exponent, mantissa, bias, is_signed);
      }))
      .def_property("mantissa", [](const P& s) { return s->mantissa_; })
      .def_property("exponent", [](const P& s) { return s->exponent_; })
      .def_property("bias", [](const P& s) { return s->bias_; })
      .def_property("signed", [](const P& s) { return s->signed_; })
      .def_property("size_bits", [](const P& s) { return s->size_bits(); })
      .def("is_signed", [](const P& s) { return s->is_signed(); })
      .def("is_integer", [](const P& s) { return s->is_integer(); })
      .def("is_floating_point", [](const P& s) { return s->is_floating_point(); })
      .def("is_ieee_754", [](const P& s) { return s->is_ieee_754(); })
      .def("has_nans", [](const P& s) { return s->has_nans(); })
      .def("has_infs", [](const P& s) { return s->has_infs(); })
      .def("has_bias", [](const P& s) { return s->has_bias(); })
      .def("max", [](const P& s) { return s->max_value(); })
      .def("min", [](const P& s) { return s->min_value(); })
      .def("__len__", [](const P&) -> int64_t { throw c10::TypeError({__func__, __FILE__, (uint32_t)__LINE__}, "ScalarType has no len()"); })
      .def("__str__", [](const P& s) { return s->name(); })
      .def("__repr__", [](const P& s) { return "ScalarType." + s->name(); })
      .def("__eq__", [](const P& a, const P& b) { return a->same(*b); })
      .def("__obj_flatten__", [](const P& s) { return std::make_tuple(std::make_tuple(std::string("ScalarType"), s->id())); })
      .def_static("__obj_unflatten__", [](std::tuple<std::tuple<std::string, int64_t>> flat) { return S::from_id(std::get<1>(std::get<0>(flat))); })
      .def_static("int_", &S::make_int)
      .def_static("uint", &S::make_uint)
      .def_static("float_IEEE754", &S::make_float_ieee)
      .def_static("float_", &S::make_float);
}

// TEST INFRASTRUCTURE ONLY — force-included (-include) when the reference's
// translation units are compiled in place by oracle/Makefile.
//
// The reference declares  `explicit X(const Settings& s = Settings())`  inside
// the class that also defines the nested `Settings` (engine/bdpt.hpp:130,
// engine/spectral/spectral_renderer.hpp:231).  MSVC accepts that; GCC and
// clang need the nested class's default member initialisers before the outer
// class is complete and reject it.  The Makefile passes
//   -D'Settings()=::oracle_ref_detail::make_default<Settings>()'
// so the default argument becomes a call to this function template, whose
// instantiation is deferred until the outer class is complete.  Semantics are
// unchanged (a value-initialised Settings); no reference file is copied or
// edited.
#pragma once
namespace oracle_ref_detail {
template <class T> T make_default() { return T(); }
}

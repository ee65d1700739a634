// ORACLE BUILD SHIM: BOOST_FOREACH as a range-for.
#pragma once
#define BOOST_FOREACH(a, b) for (a : b)

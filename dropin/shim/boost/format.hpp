// ORACLE BUILD SHIM: included by the reference but unused.
#pragma once

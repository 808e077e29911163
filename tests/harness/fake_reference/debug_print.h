// TEST HARNESS (tests/ only): stand-in for the reference's print macros when its tree is not on the include path.
#pragma once
#include <iostream>
#define DEBUG_PRINT_OUT(x) (std::cout << x << std::endl)
#define DEBUG_PRINT_ERR(x) (std::cerr << x << std::endl)

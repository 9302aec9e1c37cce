extern "C" const char* cmp_build_key(void) { return "f821bbb4f913d4535cce7c12464e4ad4c2cc1bb9249f4771ca424554dddc2700"; }

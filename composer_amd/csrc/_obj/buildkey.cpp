extern "C" const char* cmp_build_key(void) { return "65bd30286900208e8261b28e4328bc4ab265d8edd42654d5ab0132e92d0e817d"; }

# scratch: VARIANTS = {name: [(file, old, new), ...]} for tools/lab_lib.py (patched copies of csrc/, timing experiments only)
VARIANTS = {
}

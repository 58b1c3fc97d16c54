VARIANTS = {
    "nopers": [("swin.hip", "if (C == 96 && nwg >= 2048) {", "if (C == 96 && nwg >= (1ll << 40)) {")],
}

VARIANTS = {
    "noysplit": [("gemm.hip", "        if (segs[i].m_live && nsplit < 4) nsplit = npanels < 4 ? npanels : 4;", "        if (false && segs[i].m_live && nsplit < 4) nsplit = npanels < 4 ? npanels : 4;")],
}

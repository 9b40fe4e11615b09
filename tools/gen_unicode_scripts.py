"""Generates openvino_tokenizers_amd/csrc/unicode_scripts.inc: the Script (sc) and Script_Extensions (scx) properties of Unicode 16.0 as
range lists, for the regex compiler's \\p{Han} / \\p{sc:Han} / \\p{scx:Han} (csrc/regex_compile.cpp).

Why: the reference hands every pattern to PCRE2 10.46 (src/utils.cpp:256-272, src/CMakeLists.txt:185-189), and tokenizers for CJK-heavy
models write script properties (\\p{Han}, \\p{Hiragana}, \\p{Katakana}).  PCRE2 since 10.40 reads a bare script name as Script_Extensions
(`\\p{Han}` = `\\p{scx:Han}`), `\\p{sc:Han}` as Script.

Offline sources (there is no network and no UCD file in the image):
  * sc:  the Oniguruma engine inside HF `tokenizers` (Unicode 16.0.0 -- tools/gen_unicode_tables.py checks the version), `\\p{Han}` there
         is the Script property;
  * scx: Python `regex` (its Unicode version is printed; 17.0 here), `\\p{Script_Extensions=Han}`, restricted to the code points
         Unicode 16.0 assigns.  Cross-check asserted below: `regex`'s Script equals Oniguruma's on every code point 16.0 assigns, for
         every script -- so the newer data agrees with 16.0 wherever 16.0 says anything about Script; Script_Extensions is taken on that
         footing (a code point's scx may in principle differ between 16.0 and 17.0: none is known to; DESIGN.md 5 says so).
Usage: python -m tools.gen_unicode_scripts
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from tools.gen_unicode_tables import NCP, general_categories, onig_mask, GC_NAMES

OUT = Path(__file__).resolve().parent.parent / "openvino_tokenizers_amd" / "csrc" / "unicode_scripts.inc"

# Unicode 16.0 scripts: (long name, ISO 15924 code)
SCRIPTS = [
    ("Adlam", "Adlm"), ("Ahom", "Ahom"), ("Anatolian_Hieroglyphs", "Hluw"), ("Arabic", "Arab"), ("Armenian", "Armn"), ("Avestan", "Avst"),
    ("Balinese", "Bali"), ("Bamum", "Bamu"), ("Bassa_Vah", "Bass"), ("Batak", "Batk"), ("Bengali", "Beng"), ("Bhaiksuki", "Bhks"),
    ("Bopomofo", "Bopo"), ("Brahmi", "Brah"), ("Braille", "Brai"), ("Buginese", "Bugi"), ("Buhid", "Buhd"), ("Canadian_Aboriginal", "Cans"),
    ("Carian", "Cari"), ("Caucasian_Albanian", "Aghb"), ("Chakma", "Cakm"), ("Cham", "Cham"), ("Cherokee", "Cher"), ("Chorasmian", "Chrs"),
    ("Common", "Zyyy"), ("Coptic", "Copt"), ("Cuneiform", "Xsux"), ("Cypriot", "Cprt"), ("Cypro_Minoan", "Cpmn"), ("Cyrillic", "Cyrl"),
    ("Deseret", "Dsrt"), ("Devanagari", "Deva"), ("Dives_Akuru", "Diak"), ("Dogra", "Dogr"), ("Duployan", "Dupl"),
    ("Egyptian_Hieroglyphs", "Egyp"), ("Elbasan", "Elba"), ("Elymaic", "Elym"), ("Ethiopic", "Ethi"), ("Garay", "Gara"), ("Georgian", "Geor"),
    ("Glagolitic", "Glag"), ("Gothic", "Goth"), ("Grantha", "Gran"), ("Greek", "Grek"), ("Gujarati", "Gujr"), ("Gunjala_Gondi", "Gong"),
    ("Gurmukhi", "Guru"), ("Gurung_Khema", "Gukh"), ("Han", "Hani"), ("Hangul", "Hang"), ("Hanifi_Rohingya", "Rohg"), ("Hanunoo", "Hano"),
    ("Hatran", "Hatr"), ("Hebrew", "Hebr"), ("Hiragana", "Hira"), ("Imperial_Aramaic", "Armi"), ("Inherited", "Zinh"),
    ("Inscriptional_Pahlavi", "Phli"), ("Inscriptional_Parthian", "Prti"), ("Javanese", "Java"), ("Kaithi", "Kthi"), ("Kannada", "Knda"),
    ("Katakana", "Kana"), ("Kawi", "Kawi"), ("Kayah_Li", "Kali"), ("Kharoshthi", "Khar"), ("Khitan_Small_Script", "Kits"), ("Khmer", "Khmr"),
    ("Khojki", "Khoj"), ("Khudawadi", "Sind"), ("Kirat_Rai", "Krai"), ("Lao", "Laoo"), ("Latin", "Latn"), ("Lepcha", "Lepc"), ("Limbu", "Limb"),
    ("Linear_A", "Lina"), ("Linear_B", "Linb"), ("Lisu", "Lisu"), ("Lycian", "Lyci"), ("Lydian", "Lydi"), ("Mahajani", "Mahj"),
    ("Makasar", "Maka"), ("Malayalam", "Mlym"), ("Mandaic", "Mand"), ("Manichaean", "Mani"), ("Marchen", "Marc"), ("Masaram_Gondi", "Gonm"),
    ("Medefaidrin", "Medf"), ("Meetei_Mayek", "Mtei"), ("Mende_Kikakui", "Mend"), ("Meroitic_Cursive", "Merc"),
    ("Meroitic_Hieroglyphs", "Mero"), ("Miao", "Plrd"), ("Modi", "Modi"), ("Mongolian", "Mong"), ("Mro", "Mroo"), ("Multani", "Mult"),
    ("Myanmar", "Mymr"), ("Nabataean", "Nbat"), ("Nag_Mundari", "Nagm"), ("Nandinagari", "Nand"), ("New_Tai_Lue", "Talu"), ("Newa", "Newa"),
    ("Nko", "Nkoo"), ("Nushu", "Nshu"), ("Nyiakeng_Puachue_Hmong", "Hmnp"), ("Ogham", "Ogam"), ("Ol_Chiki", "Olck"), ("Ol_Onal", "Onao"),
    ("Old_Hungarian", "Hung"), ("Old_Italic", "Ital"), ("Old_North_Arabian", "Narb"), ("Old_Permic", "Perm"), ("Old_Persian", "Xpeo"),
    ("Old_Sogdian", "Sogo"), ("Old_South_Arabian", "Sarb"), ("Old_Turkic", "Orkh"), ("Old_Uyghur", "Ougr"), ("Oriya", "Orya"),
    ("Osage", "Osge"), ("Osmanya", "Osma"), ("Pahawh_Hmong", "Hmng"), ("Palmyrene", "Palm"), ("Pau_Cin_Hau", "Pauc"), ("Phags_Pa", "Phag"),
    ("Phoenician", "Phnx"), ("Psalter_Pahlavi", "Phlp"), ("Rejang", "Rjng"), ("Runic", "Runr"), ("Samaritan", "Samr"), ("Saurashtra", "Saur"),
    ("Sharada", "Shrd"), ("Shavian", "Shaw"), ("Siddham", "Sidd"), ("SignWriting", "Sgnw"), ("Sinhala", "Sinh"), ("Sogdian", "Sogd"),
    ("Sora_Sompeng", "Sora"), ("Soyombo", "Soyo"), ("Sundanese", "Sund"), ("Sunuwar", "Sunu"), ("Syloti_Nagri", "Sylo"), ("Syriac", "Syrc"),
    ("Tagalog", "Tglg"), ("Tagbanwa", "Tagb"), ("Tai_Le", "Tale"), ("Tai_Tham", "Lana"), ("Tai_Viet", "Tavt"), ("Takri", "Takr"),
    ("Tamil", "Taml"), ("Tangsa", "Tnsa"), ("Tangut", "Tang"), ("Telugu", "Telu"), ("Thaana", "Thaa"), ("Thai", "Thai"), ("Tibetan", "Tibt"),
    ("Tifinagh", "Tfng"), ("Tirhuta", "Tirh"), ("Todhri", "Todr"), ("Toto", "Toto"), ("Tulu_Tigalari", "Tutg"), ("Ugaritic", "Ugar"),
    ("Vai", "Vaii"), ("Vithkuqi", "Vith"), ("Wancho", "Wcho"), ("Warang_Citi", "Wara"), ("Yezidi", "Yezi"), ("Yi", "Yiii"),
    ("Zanabazar_Square", "Zanb"),
]


def regex_mask(prop: str, name: str) -> np.ndarray:
    import regex
    rx = regex.compile(r"\p{%s=%s}" % (prop, name))
    mask = np.zeros(NCP, dtype=bool)
    for cp in range(NCP):
        if 0xD800 <= cp < 0xE000:
            continue
        if rx.match(chr(cp)):
            mask[cp] = True
    return mask


def ranges(mask: np.ndarray):
    d = np.diff(np.concatenate([[0], mask.astype(np.int8), [0]]))
    return list(zip(np.flatnonzero(d == 1).tolist(), np.flatnonzero(d == -1).tolist()))   # [start, end)


def main():
    import regex
    gc = general_categories()
    assigned = gc != GC_NAMES.index("Cn")
    assigned[0xD800:0xE000] = False   # (surrogates have no script worth the name: Unknown)
    covered = np.zeros(NCP, dtype=bool)
    sc_r, scx_r = [], []
    for long_name, code in SCRIPTS:
        sc = onig_mask(r"\p{" + long_name + "}")
        assert sc.any(), f"Oniguruma (Unicode 16.0) has no script {long_name}"
        assert not (sc & covered).any(), f"{long_name} overlaps an earlier script"
        covered |= sc
        sc_new = regex_mask("Script", long_name) & assigned
        diff = np.flatnonzero(sc_new != (sc & assigned))
        assert diff.size == 0, f"Script={long_name}: `regex` and Oniguruma 16.0 differ on {[hex(int(x)) for x in diff[:8]]}"
        scx = regex_mask("Script_Extensions", long_name) & assigned
        if long_name not in ("Common", "Inherited"):
            assert not (sc & assigned & ~scx).any(), f"scx({long_name}) does not contain sc({long_name})"
        sc_r.append(ranges(sc))
        scx_r.append(ranges(scx))
    stray = assigned & ~covered & (gc != GC_NAMES.index("Co"))   # (private use: Script=Unknown)
    assert not stray.any(), f"assigned code points without a script: {[hex(int(x)) for x in np.flatnonzero(stray)[:8]]}"
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_unicode_scripts.py -- do not edit.\n")
        f.write(f"// Script (Oniguruma of HF tokenizers, Unicode 16.0.0) and Script_Extensions (Python regex {regex.__version__}, restricted to the\n"
                "// code points Unicode 16.0 assigns; its Script agrees with 16.0's on every one of them) as range lists [start, end).\n")
        f.write(f"static const unsigned kScriptCount = {len(SCRIPTS)};\n")
        f.write("static const char* const kScriptNames[][2] = {\n")
        for long_name, code in SCRIPTS:
            f.write(f'    {{"{long_name}", "{code}"}},\n')
        f.write("};\n")
        for tag, lists in (("Sc", sc_r), ("Scx", scx_r)):
            flat = [x for rs in lists for r in rs for x in r]
            offs = np.concatenate([[0], np.cumsum([len(rs) for rs in lists])]).tolist()
            f.write(f"static const unsigned k{tag}Offsets[{len(offs)}] = {{{','.join(map(str, offs))}}};   // script i: ranges [k{tag}Offsets[i], k{tag}Offsets[i + 1])\n")
            f.write(f"static const unsigned k{tag}Ranges[{len(flat)}] = {{\n")
            for i in range(0, len(flat), 24):
                f.write(",".join(map(str, flat[i:i + 24])) + ",\n")
            f.write("};\n")
    print(f"{len(SCRIPTS)} scripts, {sum(map(len, sc_r))} sc ranges, {sum(map(len, scx_r))} scx ranges -> {OUT}")


if __name__ == "__main__":
    main()

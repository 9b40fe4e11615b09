"""Known-answer vectors of the reference's own op-level tests (VALUES only, transcribed from
/root/reference/tests/layer_tests.py).  Patterns are the strings its pipeline steps pass to the ops
(python/openvino_tokenizers/tokenizer_pipeline.py:388-457).

  REGEX_SPLIT_KATS   tests/layer_tests.py:331-389   (input, expected pieces, pattern, behaviour, invert)
  RAGGED_TO_DENSE_KATS tests/layer_tests.py:497-573
"""
import re

WHITESPACE = (r"\w+|[^\w\s]+", "remove", True)          # RegexSplitStep.whitespace_splitter()
BERT_WHITESPACE = (r"\s+", "remove", False)             # bert_whitespace_splitter()
SPLIT_BY_CHARS = (".", "isolate", False)                # split_by_chars()
METASPACE_NEXT = ("▁", "mergedwithnext", False)
METASPACE_PREV = ("▁", "mergedwithprevious", False)
BYTE_LEVEL = (r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+", "isolate", False)
BYTE_LEVEL_DIGITS = (r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+", "isolate", False)

# The reference builds the expected CLIP pieces with Python's `re` on the same (doubly escaped) pattern.
CLIP_PATTERN = r"<\\|startoftext\\|>|<\\|endoftext\\|>|'s|'t|'re|'ve|'m|'ll|'d|[\\p{L}]+|[\\p{N}]|[^\\s\\p{L}\\p{N}]+"
CLIP = (CLIP_PATTERN, "remove", True)
TEXT2IMAGE_PROMPTS = [
    "Cinematic, a vibrant Mid-century modern dining area, colorful chairs and a sideboard, ultra realistic, many detail",
    "colibri flying near a flower, side view, forest background, natural light, photorealistic, 4k",
    "Illustration of an astronaut sitting in outer space, moon behind him",
    "A vintage illustration of a retro computer, vaporwave aesthetic, light pink and light blue",
    "A view from beautiful alien planet, very beautiful, surealism, retro astronaut on the first plane, 8k photo",
    "red car in snowy forest, epic vista, beautiful landscape, 4k, 8k",
    "A raccoon trapped inside a glass jar full of colorful candies, the background is steamy with vivid colors",
    "cute cat 4k, high-res, masterpiece, best quality, soft lighting, dynamic angle",
    "A cat holding a sign that says hello OpenVINO",
    "A small cactus with a happy face in the Sahara desert.",
]

REGEX_SPLIT_KATS = [
    ("Hello world!", ("Hello", "world", "!"), WHITESPACE),
    ("Hello     world!", ("Hello", "world!"), BERT_WHITESPACE),
    ("", ("",), WHITESPACE),
    *[(p, tuple(re.compile(CLIP_PATTERN).findall(p)), CLIP) for p in TEXT2IMAGE_PROMPTS],
    ("▁one▁two▁three▁", ("▁one", "▁two", "▁three", "▁"), METASPACE_NEXT),
    ("▁", ("▁",), METASPACE_NEXT),
    ("No split pattern", ("No split pattern",), METASPACE_NEXT),
    ("▁one▁two▁three▁", ("▁", "one▁", "two▁", "three▁"), METASPACE_PREV),
    ("▁", ("▁",), METASPACE_PREV),
    ("No split pattern", ("No split pattern",), METASPACE_PREV),
    ("split", tuple("split"), SPLIT_BY_CHARS),
    ("split by chars", tuple("split by chars"), SPLIT_BY_CHARS),
    ("Hello world!", ("Hello", " world", "!"), BYTE_LEVEL),
    ("test's great", ("test", "'s", " great"), BYTE_LEVEL),
    ("don't stop", ("don", "'t", " stop"), BYTE_LEVEL),
    ("hello 123", ("hello", " 123"), BYTE_LEVEL),
    ("Eng, but with d1gits: 123", ("Eng", ",", " but", " with", " d", "1", "gits", ":", " 123"), BYTE_LEVEL),
    ("a  b", ("a", " ", " b"), BYTE_LEVEL),
    ("Hello world!", ("Hello", " world", "!"), BYTE_LEVEL_DIGITS),
    ("hello 123", ("hello", " ", "1", "2", "3"), BYTE_LEVEL_DIGITS),
    ("Eng, but with d1gits: 123", ("Eng", ",", " but", " with", " d", "1", "gits", ":", " ", "1", "2", "3"),
     BYTE_LEVEL_DIGITS),
    ("If I have 100 million dollars?", ("If", " I", " have", " ", "1", "0", "0", " million", " dollars", "?"),
     BYTE_LEVEL_DIGITS),
    ("a1b2c3", ("a", "1", "b", "2", "c", "3"), BYTE_LEVEL_DIGITS),
    ("test 0987654321 end", ("test", " ", "0", "9", "8", "7", "6", "5", "4", "3", "2", "1", " end"), BYTE_LEVEL_DIGITS),
]

_R2D = dict(begins=[0, 3], ends=[3, 8], data=[10, 20, 100, 30, 40, 50, 200, 300], value=42)
RAGGED_TO_DENSE_KATS = [
    # (inputs, attribute pad_right, optional input pad_right, expected)
    (dict(_R2D, padding_size=10), True, None,
     [[10, 20, 100, 42, 42, 42, 42, 42, 42, 42], [30, 40, 50, 200, 300, 42, 42, 42, 42, 42]]),
    (dict(_R2D, padding_size=10), False, None,
     [[42, 42, 42, 42, 42, 42, 42, 10, 20, 100], [42, 42, 42, 42, 42, 30, 40, 50, 200, 300]]),
    (dict(_R2D, padding_size=2), True, None, [[10, 20], [30, 40]]),
    (dict(_R2D, padding_size=10), True, False,
     [[42, 42, 42, 42, 42, 42, 42, 10, 20, 100], [42, 42, 42, 42, 42, 30, 40, 50, 200, 300]]),
    (dict(_R2D, padding_size=10), False, True,
     [[10, 20, 100, 42, 42, 42, 42, 42, 42, 42], [30, 40, 50, 200, 300, 42, 42, 42, 42, 42]]),
]


# tests/layer_tests.py:405-457 (SpecialTokensSplit): (tokens [(text, strip_left, strip_right)], text, pieces, skips)
SPECIAL_TOKENS_KATS = [
    ([("<｜begin▁of▁sentence｜>", False, False)], "<｜begin▁of▁sentence｜> the user's <</SYS>>",
     ("<｜begin▁of▁sentence｜>", " the user's <</SYS>>"), [1, 0]),
    ([("<｜begin▁of▁sentence｜>", False, True)], "<｜begin▁of▁sentence｜>   the user's <</SYS>>",
     ("<｜begin▁of▁sentence｜>", "the user's <</SYS>>"), [1, 0]),
    ([("<|eot_id|>", True, False)], "    the user's <</SYS>>    <|eot_id|>", ("    the user's <</SYS>>", "<|eot_id|>"), [0, 1]),
    ([("    ", False, False)], "    def", ("    ", "def"), [1, 0]),
    ([("    ", False, False)], "    def  ", ("    ", "def  "), [1, 0]),
    ([("    ", False, False)], "    def    ", ("    ", "def", "    "), [1, 0, 1]),
    ([("def", True, False)], "_    def  _", ("_", "def", "  _"), [0, 1, 0]),
    ([("def", False, True)], "_    def  _", ("_    ", "def", "_"), [0, 1, 0]),
    ([("def", True, True)], "_    def  _def", ("_", "def", "_", "def"), [0, 1, 0, 1]),
    ([("def", True, True)], "def_    def  _def", ("def", "_", "def", "_", "def"), [1, 0, 1, 0, 1]),
    ([("def", True, True)], "defdef_    def  _def", ("def", "def", "_", "def", "_", "def"), [1, 1, 0, 1, 0, 1]),
]

# tests/layer_tests.py:601-644 (CombineSegments): (segments, expected begins/ends/data); segment ids = arange(n)
COMBINE_SEGMENTS_KATS = [
    ([{"begins": [0, 2], "ends": [2, 5], "data": [10, 20, 30, 40, 50]}, {"begins": [0, 1], "ends": [1, 3], "data": [100, 200, 300]}],
     {"begins": [0, 3], "ends": [3, 8], "data": [10, 20, 100, 30, 40, 50, 200, 300]}),
    ([{"begins": [0, 2], "ends": [2, 5], "data": [10, 20, 30, 40, 50]}, {"begins": [0, 1], "ends": [1, 3], "data": [100, 200, 300]},
      {"begins": [0, 2], "ends": [2, 3], "data": [1000, 2000, 3000]}],
     {"begins": [0, 5], "ends": [5, 11], "data": [10, 20, 100, 1000, 2000, 30, 40, 50, 200, 300, 3000]}),
]

# tests/layer_tests.py:84-139 (UTF8Validate): expected = bytes.decode(errors="ignore" | "replace")
UTF8_VALIDATE_KATS = [
    b"Eng... test, string?!",
    b"\xe2\x82\xac",
    "Проверка, как работает кириллица Љ љ Ђ ђ".encode(),
    "測試字符串".encode(),
    "Tester, la chaîne...".encode(),
    "سلسلة الاختبار".encode(),
    "מחרוזת בדיקה".encode(),
    "Сынақ жолы á".encode(),
    "😁😁".encode(),
    "🤣🤣🤣😁😁😁😁".encode(),
    "🫠".encode(),
    "介绍下清华大学".encode(),
    "折纸的过程看似简单，其实想要做好，还是需要一套很复杂的工艺。以折一支玫瑰花为例，我们可以将整个折纸过程分成三个阶段，即：创建栅格折痕，制作立体基座，完成花瓣修饰。".encode(),
    b"\x81First byte is invalid utf8",
    b"\x80\x80\x80",
    bytes([0b11000000, 0b11000000, 0b11000000]),
    bytes([0b11110000, 0b10010011, 0b10000001, 0b11101000, 0b11110000, 0b10010011, 0b10000001, 0b10101000]),
    bytes([0b11110000, 0b10011111, 0b10011000, 0b11000001, 0b11110000, 0b10011111, 0b10011000, 0b10000001]),
    b"\xc0\x80",
    b"\xe0\x81\x81",
    b"\xf0\x80\x80\x80",
    b"\xe2\x28\xa1",
    b"the following block is invalid \xe2\x28\xa1 but this text is valid",
    b"A\xc3\x28B",
    b"\xe2\x82",
    b"A\xc3\xa9\xe2\x82\xac\xf0\x90\x8d\x88",
]

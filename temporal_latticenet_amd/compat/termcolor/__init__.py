"""harness stand-in for termcolor.colored (models.py:4, 246-248)"""


def colored(text, *args, **kwargs):
    return text

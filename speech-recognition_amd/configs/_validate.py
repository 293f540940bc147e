"""Field validation for the config dataclasses.

The reference declares its configs as pydantic-v1 dataclasses (positional-argument errors are
TypeError from the generated __init__, type errors are pydantic.ValidationError).  This image has
pydantic 2.x, so the configs are standard dataclasses whose __post_init__ validates (and coerces)
every field through pydantic TypeAdapters, raising the same exception types."""
import dataclasses
import typing

from pydantic import TypeAdapter, ValidationError
from pydantic_core import InitErrorDetails, PydanticCustomError

REQUIRED = object()   # sentinel default for fields the reference marks `Field(...)`


def validate_fields(obj, skip=()):
    hints = typing.get_type_hints(type(obj))
    errors = []
    for f in dataclasses.fields(obj):
        if f.name in skip:
            continue
        v = getattr(obj, f.name)
        if v is REQUIRED:
            errors.append(InitErrorDetails(type="missing", loc=(f.name,), input=None))
            continue
        try:
            setattr(obj, f.name, TypeAdapter(hints[f.name]).validate_python(v))
        except ValidationError as e:
            for err in e.errors():
                errors.append(InitErrorDetails(type=PydanticCustomError(err["type"], err["msg"]), loc=(f.name,) + tuple(err["loc"]),
                                               input=err.get("input")))
    if errors:
        raise ValidationError.from_exception_data(type(obj).__name__, errors)

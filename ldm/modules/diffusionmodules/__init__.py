"""Reference-path shim package: re-exports the MI355X-native implementations under the reference dotted paths."""
